"""Drop-in for the reference's ``CodeGenerator`` (reference sr/models.py:125-225).

Same construction (``CodeGenerator(h)`` from the vocoder ``config.json``), same
checkpoint format (``load_state_dict(ckpt['generator'])`` with the 293
weight_g/weight_v/bias keys), same call (``generator(code=, f0=, spkr=)`` ->
``[B,1,320*T]``) -- but the forward is one call into libdissc_hip.so.

Beyond the reference: ``lengths=`` (i32 [B]) runs a ragged batch with every
utterance sample-exact to a B=1 run of its own (the reference cannot batch).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import lib, check


class AttrDict(dict):
    """Config dict with attribute access (reference sr/utils.py:77-80)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


def fold_weight_norm(weight_g, weight_v):
    """remove_weight_norm(): w = g * v/||v|| over all dims but 0 (reference
    sr/models.py:43-47,116-122).  torch._weight_norm is the primitive the
    reference's hook evaluates, so the folded weights are bit-identical."""
    return torch._weight_norm(weight_v, weight_g, 0)


class CodeGenerator:
    """HiFi-GAN generator conditioned on units + F0 + speaker, on one MI355X."""

    _UNSUPPORTED = ("lambda_commit", "lambda_commit_code", "f0_quantizer_path")

    _PRECISIONS = {"fp32": 0, "split_bf16": 1}

    def __init__(self, h, precision=None):
        """h: the vocoder config (reference sr/configs/*.json).  precision (not in the reference):
        None = the process-wide "precision" option (default exact fp32), "fp32", or "split_bf16"
        (bf16 matrix cores with hi/lo operand split and fp32 accumulation, waveform RMS ~4e-6)."""
        if precision is not None and precision not in self._PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(self._PRECISIONS)} or None")
        self.precision = precision
        self.h = AttrDict(h)
        for k in self._UNSUPPORTED:
            if self.h.get(k, None):
                # VQ-VAE branches (reference sr/models.py:137-156) are not part of the
                # shipped DISSC configs.
                raise NotImplementedError(f"config option '{k}' (VQ-VAE branch) is not supported")
        if str(self.h.get("resblock", "1")) != "1":
            raise NotImplementedError("only resblock type '1' is supported")
        self.num_kernels = len(self.h.resblock_kernel_sizes)
        self.num_upsamples = len(self.h.upsample_rates)
        self.f0 = self.h.get("f0", None)
        self.multispkr = self.h.get("multispkr", None)
        self.device = None
        self._handle = None
        self._raw = None       # state dict as loaded (weight_g / weight_v)
        self._folded = None
        self._ws = None
        self.training = False

    # -- nn.Module-like surface used by the reference's inference code ----------------
    def load_state_dict(self, sd, strict=True):
        """Accepts the reference checkpoint layout (weight_g/weight_v per conv) or an
        already folded one (``.weight``)."""
        got = set(sd.keys())
        missing, known = [], set()
        for name in self._conv_names():
            known.update({name + ".bias", name + ".weight", name + ".weight_g", name + ".weight_v"})
            if name + ".bias" not in got:
                missing.append(name + ".bias")
            if name + ".weight" not in got and not {name + ".weight_g", name + ".weight_v"} <= got:
                missing.append(name + ".weight_g")
        for name in ["dict.weight"] + (["spkr.weight"] if self.multispkr else []):
            known.add(name)
            if name not in got:
                missing.append(name)
        unexpected = sorted(got - known)
        if missing or (strict and unexpected):
            raise RuntimeError("Error(s) in loading state_dict for CodeGenerator: "
                               f"missing {missing[:5]}, unexpected {unexpected[:5]}")
        self._raw = {k: v.detach().to("cpu", torch.float32) for k, v in sd.items() if k in known}
        self._folded = None
        self._destroy()
        return self

    def eval(self):
        self.training = False
        return self

    def to(self, device):
        if isinstance(device, int):
            device = f"cuda:{device}"
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.DisscError("dissc_amd.CodeGenerator runs on an MI355X only (device='cuda:N')")
        return self

    def cuda(self, device=0):
        return self.to(device)

    def remove_weight_norm(self):
        """Fold weight_g/weight_v pairs (idempotent; also done lazily on first call)."""
        if self._raw is None:
            raise RuntimeError("load_state_dict() first")
        out = {}
        for k, v in self._raw.items():
            if k.endswith(".weight_g"):
                base = k[: -len(".weight_g")]
                out[base + ".weight"] = fold_weight_norm(v, self._raw[base + ".weight_v"])
            elif not k.endswith(".weight_v"):
                out[k] = v
        self._folded = out
        return self

    def _conv_names(self):
        names = ["conv_pre"] + [f"ups.{i}" for i in range(self.num_upsamples)]
        for i in range(self.num_upsamples * self.num_kernels):
            for grp in ("convs1", "convs2"):
                names += [f"resblocks.{i}.{grp}.{m}" for m in range(3)]
        return names + ["conv_post"]

    # -- native handle ----------------------------------------------------------------
    def _config(self):
        h = self.h
        c = _lib.DisscGenConfig()
        c.model_in_dim = int(h.get("model_in_dim", 128))
        c.upsample_initial_channel = int(h.upsample_initial_channel)
        c.num_upsamples = self.num_upsamples
        for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
            c.upsample_rates[i] = int(u)
            c.upsample_kernel_sizes[i] = int(k)
        c.num_kernels = self.num_kernels
        for j, (k, ds) in enumerate(zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes)):
            c.resblock_kernel_sizes[j] = int(k)
            if len(ds) != 3:
                raise NotImplementedError("ResBlock1 needs 3 dilations per kernel size")
            for m, d in enumerate(ds):
                c.resblock_dilations[j][m] = int(d)
        c.num_embeddings = int(h.num_embeddings)
        c.embedding_dim = int(h.embedding_dim)
        c.num_speakers = 200  # nn.Embedding(200, ...) reference sr/models.py:133
        c.has_f0 = 1 if self.f0 else 0
        c.has_spkr = 1 if self.multispkr else 0
        return c

    def prepare(self):
        """Build the native handle now (weight-norm fold, host packing of the direct and transform-domain weights,
        upload) instead of lazily on the first call -- so that a caller can account for it (tools/cli_wall.py)."""
        self._ensure()
        return self

    def _ensure(self):
        if self._handle is not None:
            return
        if self.device is None:
            self.to("cuda:0")
        if self._folded is None:
            self.remove_weight_norm()
        with torch.cuda.device(self.device):
            cfg = self._config()
            table, keep = _lib.make_tensor_table(self._folded)
            hnd = ctypes.c_void_p()
            # the handle's arithmetic is an argument of its constructor: no process-wide state is touched
            prec = -1 if self.precision is None else self._PRECISIONS[self.precision]
            check(lib.dissc_gen_create_ex(ctypes.byref(cfg), table, len(keep), prec, ctypes.byref(hnd)),
                  "dissc_gen_create_ex")
            self._handle = hnd
            self.hop = lib.dissc_gen_hop(hnd)

    def _destroy(self):
        if self._handle is not None:
            lib.dissc_gen_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def flops(self, frames):
        """algorithmic FLOPs (2 x multiply-adds of the direct form) for `frames` code frames"""
        self._ensure()
        return lib.dissc_gen_flops(self._handle, int(frames))

    def flops_executed(self, frames):
        """FLOPs the matrix pipe executes: fewer than flops() where ResBlock convs run in the Toom-Cook F(4,3)
        transform domain (csrc/conv_wino.hip)"""
        self._ensure()
        return lib.dissc_gen_flops_executed(self._handle, int(frames))

    def _workspace(self, B, T):
        need = lib.dissc_gen_workspace_bytes(self._handle, B, T)
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.device:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws, need

    # -- forward ----------------------------------------------------------------------
    @staticmethod
    def _upsample(signal, max_frames):
        """Integer repeat of a conditioning signal (reference sr/models.py:158-177)."""
        if signal.dim() == 2:
            signal = signal.unsqueeze(2)
        elif signal.dim() != 3:
            signal = signal.view(-1, 1, 1)
        cond = signal.shape[2]
        rep = max_frames // cond
        if (max_frames - cond * rep) // max(rep, 1) > 0:
            raise NotImplementedError(
                'Padding condition signal - misalignment between condition features.')
        return signal.repeat_interleave(rep, dim=2)

    def validate_host_ids(self, code=None, spkr=None):
        """IndexError for unit / speaker ids outside the embedding tables, like nn.Embedding in the reference
        (host arrays or tensors; callers that upload ids themselves, e.g. dissc_amd.harness, call this first)."""
        for name, t, rows in (("code", code, int(self.h.num_embeddings)), ("spkr", spkr if self.multispkr else None, 200)):
            if t is None:
                continue
            t = torch.as_tensor(t)
            if t.numel():
                lo, hi = int(t.min()), int(t.max())
                if lo < 0 or hi >= rows:
                    raise IndexError(f"{name} id out of range: [{lo}, {hi}] not within [0, {rows})")

    def forward(self, **kwargs):
        self._ensure()
        extra = [k for k in kwargs if k not in ("code", "f0", "spkr", "lengths")]
        if extra:
            raise NotImplementedError(f"extra conditioning features {extra} are not supported")
        dev = self.device
        code = kwargs["code"].to(dev, torch.int64)
        if code.dim() != 2:
            raise ValueError("code must be [B,T]")
        f0 = None
        if self.f0:
            f0 = kwargs["f0"].to(dev, torch.float32)
            if f0.dim() == 2:
                f0 = f0.unsqueeze(1)
            if code.shape[-1] < f0.shape[-1]:
                code = self._upsample(code.unsqueeze(1), f0.shape[-1]).squeeze(1)
            elif f0.shape[-1] != code.shape[-1]:
                f0 = self._upsample(f0, code.shape[-1])
            if f0.shape[-1] != code.shape[-1] or f0.shape[0] != code.shape[0] or f0.shape[1] != 1:
                # the reference fails here too (torch.cat of mismatched lengths, sr/models.py:213-215);
                # the kernels index f0[b*T + t] and must never see a shorter row
                raise RuntimeError(f"Sizes of tensors must match: code {tuple(code.shape)} vs f0 {tuple(f0.shape)} "
                                   "after conditioning upsample")
            f0 = f0.reshape(code.shape[0], -1).contiguous()
        code = code.contiguous()
        B, T = code.shape
        # nn.Embedding raises IndexError on a bad id (reference sr/models.py:189,207); checked for
        # host tensors (ids read from files), device-resident ids come from our own kernels
        self.validate_host_ids(kwargs["code"] if kwargs["code"].device.type == "cpu" else None,
                               kwargs.get("spkr") if self.multispkr and kwargs["spkr"].device.type == "cpu" else None)
        spkr = None
        if self.multispkr:
            spkr = kwargs["spkr"].to(dev, torch.int64).reshape(-1).contiguous()
            if spkr.numel() != B:
                raise ValueError("spkr must be [B,1]")
        lengths = kwargs.get("lengths", None)
        if lengths is not None:
            lengths = torch.as_tensor(lengths).to(dev, torch.int32).contiguous()
            if lengths.numel() != B:
                raise ValueError("lengths must be [B]")
        with torch.cuda.device(dev):
            out = torch.empty(B, 1, self.hop * T, dtype=torch.float32, device=dev)
            ws, need = self._workspace(B, T)
            check(lib.dissc_gen_forward(
                self._handle, code.data_ptr(), f0.data_ptr() if f0 is not None else None,
                spkr.data_ptr() if spkr is not None else None,
                lengths.data_ptr() if lengths is not None else None, B, T, out.data_ptr(),
                ws.data_ptr(), need, _lib.current_stream_ptr(dev)), "dissc_gen_forward")
        return out

    __call__ = forward


def wav_postprocess_(wav, n_samples):
    """In place on the GPU: int16 truncation/wrap then peak normalisation, per utterance
    (reference sr/inference.py:73-75,206).  wav f32 [B,1,L] or [B,L]; n_samples i32 [B]."""
    w2 = wav.view(wav.shape[0], -1)
    n = torch.as_tensor(n_samples).to(wav.device, torch.int32).contiguous()
    with torch.cuda.device(wav.device):
        check(lib.dissc_wav_postprocess(w2.data_ptr(), n.data_ptr(), w2.shape[0], w2.shape[1],
                                        _lib.current_stream_ptr(wav.device)), "dissc_wav_postprocess")
    return wav
