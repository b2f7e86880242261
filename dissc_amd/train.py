"""Training of DISSC's length / pitch predictors on the MI355X: the loops of the reference's
train_len_predictor.py / train_f0_predictor.py with every optimisation step (train-mode forward, LenSumLoss /
PitchLoss, backward, Adam) as ONE call into libdissc_hip.so (csrc/train.hip, C ABI ``dissc_train_*``).

    trainer = Trainer("len", state_dict, lr=3e-4, norm=(mean, std)).to("cuda:0")
    loss = trainer.step(seq, spk_id, target)            # draws the train()-mode masks itself, or takes them
    sd = trainer.state_dict()                            # reference checkpoint layout -> dissc_amd.predictors / torch

Checkpoints keep the reference's layout (``best_model.pth`` = the module's state_dict, ``len_norm_stats.pth``).
Validation runs the inference kernels (dissc_amd.predictors) on the exported state dict.  The reference draws its
masks from the CUDA generator; here they come from a seeded torch CPU generator (or from the caller): the SAME
distribution (token embeddings zeroed where u > keep_rate; PositionalEncoding dropout p = 0.4), not the same stream.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from . import formats
from ._lib import check, lib

KINDS = {"len": 0, "new": 1, "base": 2}
MASKING_RATE = {"len": 0.2, "new": 0.4, "base": 0.4}  # reference constructors' masking_rate defaults
PE_DROPOUT = 0.4                                       # PositionalEncoding(dropout=0.4), model/pitch_predictor.py:7


def _bind():
    vp, i32 = ctypes.c_void_p, ctypes.c_int
    lib.dissc_train_create.argtypes = [i32, ctypes.POINTER(_lib.DisscTensor), ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.dissc_train_destroy.argtypes = [vp]
    lib.dissc_train_destroy.restype = None
    lib.dissc_train_set_len_norm.argtypes = [vp, ctypes.c_float, ctypes.c_float]
    lib.dissc_train_set_pitch_stats.argtypes = [vp, vp, vp, i32]
    lib.dissc_train_workspace_bytes.argtypes = [vp, i32, i32]
    lib.dissc_train_workspace_bytes.restype = ctypes.c_size_t
    lib.dissc_train_step.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, ctypes.c_float, ctypes.c_float, vp, vp,
                                     ctypes.c_size_t, vp]
    lib.dissc_train_num_tensors.argtypes = [vp]
    lib.dissc_train_tensor_name.argtypes = [vp, i32]
    lib.dissc_train_tensor_name.restype = ctypes.c_char_p
    lib.dissc_train_tensor_numel.argtypes = [vp, i32]
    lib.dissc_train_tensor_numel.restype = ctypes.c_longlong
    lib.dissc_train_steps.argtypes = [vp]
    lib.dissc_train_steps.restype = ctypes.c_longlong
    lib.dissc_train_read.argtypes = [vp, i32, i32, vp, vp]


_bind()


class Trainer:
    """One predictor under training.  kind: 'len' | 'new' | 'base'; state_dict: the reference module's state_dict
    (e.g. from ``init_state_dict``); norm: (norm_mean, norm_std) of the length labels; stats: (id2pitch_mean,
    id2pitch_std) tensors for the pitch loss."""

    def __init__(self, kind, state_dict, lr, norm=None, stats=None, seed=0):
        if kind not in KINDS:
            raise ValueError(f"kind must be one of {sorted(KINDS)}")
        self.kind, self.lr = kind, float(lr)
        self._sd0 = {k: v.detach().to("cpu").clone() for k, v in state_dict.items()}
        self.norm = (float(norm[0]), float(norm[1])) if norm is not None else (0.0, 1.0)
        self.stats = None if stats is None else tuple(torch.as_tensor(s, dtype=torch.float32).contiguous() for s in stats)
        self.device = None
        self._h = None
        self._ws = None
        self._gen = torch.Generator().manual_seed(int(seed))
        self._nbt = {k: int(v) for k, v in self._sd0.items() if k.endswith("num_batches_tracked")}

    def to(self, device):
        self.device = torch.device(f"cuda:{device}" if isinstance(device, int) else device)
        if self.device.type != "cuda":
            raise _lib.DisscError("dissc_amd.train.Trainer runs on an MI355X only")
        return self

    def _ensure(self):
        if self._h is not None:
            return
        if self.device is None:
            self.to("cuda:0")
        named = {k: v.float() for k, v in self._sd0.items() if not k.endswith("num_batches_tracked")}
        with torch.cuda.device(self.device):
            table, keep = _lib.make_tensor_table(named)
            h = ctypes.c_void_p()
            check(lib.dissc_train_create(KINDS[self.kind], table, len(keep), ctypes.byref(h)), "dissc_train_create")
            self._h = h
            if self.kind == "len":
                check(lib.dissc_train_set_len_norm(h, self.norm[0], self.norm[1]), "dissc_train_set_len_norm")
            else:
                if self.stats is None:
                    raise ValueError("the pitch loss needs stats=(id2pitch_mean, id2pitch_std)")
                m, s = self.stats
                check(lib.dissc_train_set_pitch_stats(h, m.data_ptr(), s.data_ptr(), int(m.numel())),
                      "dissc_train_set_pitch_stats")
        self._names = [lib.dissc_train_tensor_name(h, i).decode() for i in range(lib.dissc_train_num_tensors(h))]

    def __del__(self):
        try:
            if self._h is not None:
                lib.dissc_train_destroy(self._h)
        except Exception:
            pass

    # -- masks of train() mode ----------------------------------------------------------------------------
    def draw_masks(self, B, L):
        """(keep f32 [B,L], pe_mult f32 [B,L,32] | None): reference model/len_predictor.py:37-39,
        model/pitch_predictor.py:74-76 (mask = uniform > keep_rate -> zeroed) and PositionalEncoding's dropout"""
        keep_rate = 1.0 - MASKING_RATE[self.kind]
        keep = (torch.rand(B, L, generator=self._gen) <= keep_rate).float()
        pe_mult = None
        if self.kind == "new":
            pe_mult = (torch.rand(B, L, 32, generator=self._gen) >= PE_DROPOUT).float() / (1.0 - PE_DROPOUT)
        return keep, pe_mult

    def step(self, seq, spk_id, target, keep=None, pe_mult=None, pad_value=None):
        """One optimisation step on a padded batch (seq int [B,L] with pad token n_tokens, spk_id int [B,1],
        target f32 [B,L]).  Returns the summed loss as a 0-dim CUDA tensor (no host sync)."""
        self._ensure()
        dev = self.device
        seq = torch.as_tensor(seq)
        B, L = seq.shape
        if keep is None:
            keep, pm = self.draw_masks(B, L)
            pe_mult = pm if pe_mult is None else pe_mult
        pad = float(pad_value) if pad_value is not None else (-1.0 if self.kind == "len" else -100.0)
        seq = seq.to(dev, torch.int64).contiguous()
        spk = torch.as_tensor(spk_id).to(dev, torch.int64).reshape(-1).contiguous()
        tgt = torch.as_tensor(target).to(dev, torch.float32).contiguous()
        kp = torch.as_tensor(keep).to(dev, torch.float32).contiguous()
        pm = None if pe_mult is None or self.kind != "new" else torch.as_tensor(pe_mult).to(dev, torch.float32).contiguous()
        if spk.numel() != B or tuple(tgt.shape) != (B, L) or tuple(kp.shape) != (B, L):
            raise ValueError("shapes: seq [B,L], spk_id [B,1], target [B,L], keep [B,L]")
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            need = lib.dissc_train_workspace_bytes(self._h, B, L)
            if self._ws is None or self._ws.numel() < need:
                self._ws = None
                self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
            check(lib.dissc_train_step(self._h, seq.data_ptr(), spk.data_ptr(), tgt.data_ptr(), kp.data_ptr(),
                                       pm.data_ptr() if pm is not None else None, B, L, pad, self.lr, loss.data_ptr(),
                                       self._ws.data_ptr(), need, _lib.current_stream_ptr(dev)), "dissc_train_step")
        for k in self._nbt:
            self._nbt[k] += 1
        return loss[0]

    def _read(self, i, which):
        n = lib.dissc_train_tensor_numel(self._h, i)
        out = torch.empty(n, dtype=torch.float32)
        with torch.cuda.device(self.device):
            check(lib.dissc_train_read(self._h, i, which, out.data_ptr(), _lib.current_stream_ptr(self.device)),
                  "dissc_train_read")
        return out

    def state_dict(self):
        """the reference module's state_dict (same keys, shapes and order as the one given to the constructor)"""
        self._ensure()
        got = {name: self._read(i, 0) for i, name in enumerate(self._names)}
        out = {}
        for k, v in self._sd0.items():
            if k.endswith("num_batches_tracked"):
                out[k] = torch.tensor(self._nbt[k], dtype=v.dtype)
            else:
                out[k] = got[k].view(v.shape).to(v.dtype)
        return out

    def grads(self):
        """{name: gradient of the last step} for the trainable tensors"""
        self._ensure()
        skip = ("running_mean", "running_var")
        return {name: self._read(i, 1).view(self._sd0[name].shape) for i, name in enumerate(self._names)
                if not name.endswith(skip) and name != "pe.pe"}


# ---------------------------------------------------------------------------------------------------------
# initial weights and datasets (host; reference model constructors / dataset/*.py)
# ---------------------------------------------------------------------------------------------------------
def init_state_dict(kind, n_tokens=100, n_speakers=99, seed=None):
    """A freshly initialised state_dict in the reference's layout with PyTorch's default initialisers
    (nn.Embedding N(0,1) with a zero padding row, nn.Conv1d kaiming_uniform(a=sqrt 5) + uniform bias,
    BatchNorm1d ones / zeros) -- built from torch.nn modules on the host (layout glue, not compute)."""
    from torch import nn
    if seed is not None:
        torch.manual_seed(seed)
    E = 32
    mods = {"token_emb": nn.Embedding(n_tokens + 1, E, padding_idx=n_tokens)}
    if kind == "len":
        mods["spk_emb"] = nn.Embedding(n_speakers, E)
        convs = [("cnn1", 2 * E, 128, 3, "bn1")] + [(f"cnn1{i}", 128, 128, 3, f"bn1{i}") for i in range(1, 7)] + \
                [("cnn2", 128, 1, 3, None)]
    else:
        base = kind == "base"
        mods["spk_emb"] = nn.Embedding(n_speakers + 1, E, padding_idx=n_speakers)
        convs = [("cnn1", 2 * E, 128, 3, "bn1" if base else None)] + \
                [(f"cnn1{i}", 128, 128, 3, f"bn1{i}" if base else None) for i in range(1, 8)] + \
                [("cnn2", 128, 128, 3, None if base else "bn2"), ("cnn_class1", 128, 128, 3, "bn_c1" if base else None),
                 ("cnn_class2", 128, 1, 1, None), ("cnn_reg1", 128, 128, 3, "bn_r1" if base else None),
                 ("cnn_reg2", 128, 1, 1, None)]
    sd = {}
    for name in ("token_emb", "spk_emb"):
        sd[name + ".weight"] = mods[name].weight.detach().clone()
    if kind == "new":
        max_len = 850
        pe = torch.cat([torch.repeat_interleave(torch.linspace(0, 1, max_len).unsqueeze(-1), E // 2, dim=-1),
                        torch.repeat_interleave(torch.linspace(1, 0, max_len).unsqueeze(-1), E // 2, dim=-1)], dim=-1)
        sd["pe.pe"] = pe.unsqueeze(0)
    bns = {}
    for name, cin, cout, k, bn in convs:
        c = nn.Conv1d(cin, cout, k)
        sd[name + ".weight"], sd[name + ".bias"] = c.weight.detach().clone(), c.bias.detach().clone()
        if bn:
            bns[bn] = cout
    # key order of the reference modules: convs and their BatchNorms interleaved as declared; order is irrelevant to
    # load_state_dict, so the BatchNorm entries simply follow
    for bn, c in bns.items():
        sd[bn + ".weight"], sd[bn + ".bias"] = torch.ones(c), torch.zeros(c)
        sd[bn + ".running_mean"], sd[bn + ".running_var"] = torch.zeros(c), torch.ones(c)
        sd[bn + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    return sd


def _pad(seqs, value, dtype):
    L = max(len(s) for s in seqs)
    out = torch.full((len(seqs), L), value, dtype=dtype)
    for i, s in enumerate(seqs):
        out[i, :len(s)] = torch.as_tensor(s, dtype=dtype)
    return out


def load_len_dataset(path, spk_id_dict, n_tokens=100, pad_value=-1):
    """reference dataset/len_dataset.py:20-32: dedup'd units, run lengths as float labels, the WHOLE file padded to its
    longest sequence -> (vals i32 [N,L], lens f32 [N,L], spk i32 [N,1], names)"""
    vals, counts, spk, names = [], [], [], []
    for d in formats.read_manifest(path):
        v, c = [], []
        for u in d["units"]:
            if v and v[-1] == u:
                c[-1] += 1
            else:
                v.append(int(u))
                c.append(1)
        vals.append(v), counts.append(c)
        spk.append(spk_id_dict[d["audio"].split("_")[0]]), names.append(d["audio"])
    return _pad(vals, n_tokens, torch.int32), _pad(counts, pad_value, torch.float32), \
        torch.tensor(spk, dtype=torch.int32).view(-1, 1), names


def load_pitch_dataset(path, spk_id_dict, f0_param_dict, n_tokens=100, pad_value=-100):
    """reference dataset/pitch_dataset.py:23-42: units, per-speaker z-scored F0 (unvoiced frames stay 0) ->
    (vals i32 [N,L], f0 f32 [N,L], spk i32 [N,1], names)"""
    vals, fs, spk, names = [], [], [], []
    for d in formats.read_manifest(path):
        name = d["audio"].split("_")[0]
        f0 = torch.tensor(d["f0"], dtype=torch.float32)
        ii = f0 != 0
        f0[ii] -= f0_param_dict[name]["mean"]
        f0[ii] /= f0_param_dict[name]["std"]
        vals.append(d["units"]), fs.append(f0)
        spk.append(spk_id_dict[name]), names.append(d["audio"])
    return _pad(vals, n_tokens, torch.int32), _pad(fs, pad_value, torch.float32), \
        torch.tensor(spk, dtype=torch.int32).view(-1, 1), names


def batches(n, batch_size, shuffle, generator=None):
    order = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
    for i in range(0, n, batch_size):
        yield order[i:i + batch_size]


def write_log(path, split, epoch, metrics):
    """the reference logs TF summaries (utils.py:22-37, tensorflow is not available here): one JSON line per epoch"""
    import json
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "a") as f:
        f.write(json.dumps({"split": split, "epoch": epoch, **{k: float(v) for k, v in metrics.items()}}) + "\n")
