"""HuBERT-base unit encoder on the MI355X: the drop-in for what ``data/encode.py`` gets from
``textless.data.speech_encoder.SpeechEncoder`` (reference data/encode.py:21-22,32).

    enc = SpeechEncoder.from_files('hubert_base_ls960.pt', 'km100.bin', layer=6).to('cuda:0')
    out = enc(waveform)      # waveform f32 [1,N] @16 kHz
    out['units'] i64 [T], out['dense'] f32 [T,768], out['durations'] [T], out['f0'] [T]

Checkpoint formats accepted (no network: files must be local):
  * dense model: a fairseq checkpoint (``{'model': state_dict}`` or a bare state_dict with
    fairseq key names) or a HuggingFace ``HubertModel`` state_dict (key names converted);
  * k-means: a ``[K,768]`` float array as ``.npy`` / ``.pt``, or a joblib/pickle of an object
    with ``cluster_centers_`` (what textless ships).
``f0`` (YAAPT) is off the --pred_pitch path and returned as zeros.
"""
import ctypes
import os
import pickle

import numpy as np
import torch

from . import _lib
from ._lib import lib, check


def hf_to_fairseq(sd):
    """transformers.HubertModel key names -> fairseq names (inverse of HF's conversion map)."""
    out = {}
    for k, v in sd.items():
        k = k[len("hubert."):] if k.startswith("hubert.") else k
        if k.startswith("feature_extractor.conv_layers."):
            p = k.split(".")
            sub = "0" if p[3] == "conv" else "2"
            out[f"feature_extractor.conv_layers.{p[2]}.{sub}.{p[4]}"] = v
        elif k.startswith("feature_projection.projection."):
            out["post_extract_proj." + k.split(".")[-1]] = v
        elif k.startswith("feature_projection.layer_norm."):
            out["layer_norm." + k.split(".")[-1]] = v
        elif k.endswith("pos_conv_embed.conv.parametrizations.weight.original0") or k.endswith("pos_conv_embed.conv.weight_g"):
            out["encoder.pos_conv.0.weight_g"] = v
        elif k.endswith("pos_conv_embed.conv.parametrizations.weight.original1") or k.endswith("pos_conv_embed.conv.weight_v"):
            out["encoder.pos_conv.0.weight_v"] = v
        elif k.endswith("pos_conv_embed.conv.bias"):
            out["encoder.pos_conv.0.bias"] = v
        elif k.startswith("encoder.layers."):
            k2 = (k.replace("feed_forward.intermediate_dense.", "fc1.").replace("feed_forward.output_dense.", "fc2.")
                   .replace("attention.", "self_attn."))
            if ".layer_norm." in k2 and "final_layer_norm" not in k2:
                k2 = k2.replace(".layer_norm.", ".self_attn_layer_norm.")
            out[k2] = v
        elif k.startswith("encoder.layer_norm."):
            out[k] = v
    return out


def load_kmeans_centers(path):
    if path.endswith(".npy"):
        c = np.load(path)
    elif path.endswith((".pt", ".pth")):
        c = torch.load(path, map_location="cpu")
        c = c.numpy() if torch.is_tensor(c) else np.asarray(c)
    else:
        try:
            import joblib
            obj = joblib.load(path)
        except Exception:
            with open(path, "rb") as f:
                obj = pickle.load(f)
        c = obj.cluster_centers_ if hasattr(obj, "cluster_centers_") else np.asarray(obj)
    return torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32))


class HubertEncoder:
    """Dense HuBERT features (layer ``n_layers`` output) + optional k-means units."""

    def __init__(self, state_dict, centers=None, n_layers=6):
        sd = state_dict.get("model", state_dict) if isinstance(state_dict, dict) else state_dict
        if not any(k.startswith("post_extract_proj") for k in sd):
            sd = hf_to_fairseq(sd)
        self._sd = {k: v.detach().to("cpu", torch.float32) for k, v in sd.items() if torch.is_tensor(v)}
        self.n_layers = n_layers
        self.centers = None if centers is None else torch.as_tensor(centers, dtype=torch.float32).contiguous()
        self.device = None
        self._handle = None
        self._ws = None

    def to(self, device):
        if isinstance(device, int):
            device = f"cuda:{device}"
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.DisscError("dissc_amd.HubertEncoder runs on an MI355X only")
        return self

    def _tensors(self):
        sd = self._sd
        t = {}
        for k, v in sd.items():
            if k.startswith(("feature_extractor.", "layer_norm.", "post_extract_proj.", "encoder.layer_norm.")):
                t[k] = v
            elif k.startswith("encoder.layers."):
                if int(k.split(".")[2]) < self.n_layers:
                    t[k] = v
        if "encoder.pos_conv.0.weight" in sd:
            t["encoder.pos_conv.0.weight"] = sd["encoder.pos_conv.0.weight"]
        else:  # weight_norm(dim=2): fold like the module does on every forward
            t["encoder.pos_conv.0.weight"] = torch._weight_norm(sd["encoder.pos_conv.0.weight_v"],
                                                                sd["encoder.pos_conv.0.weight_g"], 2)
        t["encoder.pos_conv.0.bias"] = sd["encoder.pos_conv.0.bias"]
        return t

    def _ensure(self):
        if self._handle is not None:
            return
        if self.device is None:
            self.to("cuda:0")
        with torch.cuda.device(self.device):
            table, keep = _lib.make_tensor_table(self._tensors())
            h = ctypes.c_void_p()
            c = self.centers
            check(lib.dissc_hubert_create(self.n_layers, table, len(keep), c.data_ptr() if c is not None else None,
                                          int(c.shape[0]) if c is not None else 0, ctypes.byref(h)),
                  "dissc_hubert_create")
            self._handle = h

    def __del__(self):
        try:
            if self._handle is not None:
                lib.dissc_hubert_destroy(self._handle)
        except Exception:
            pass

    # textless' HubertFeatureReader feeds the model at most this many samples at a time and
    # concatenates the features of the chunks (100 s at 16 kHz) [3P-unverified, from memory]
    MAX_CHUNK = 1600000

    def forward(self, wav, n_samples=None, want_dense=True):
        """wav f32 [B,N] -> dict(units i64 [B,T], dense f32 [B,T,768], frames i32 [B] on the host, frames_dev: the same on the device)"""
        self._ensure()
        dev = self.device
        wav = torch.as_tensor(wav).to(dev, torch.float32)
        if wav.dim() == 1:
            wav = wav.unsqueeze(0)
        if wav.shape[1] > self.MAX_CHUNK:
            return self._forward_chunked(wav, n_samples, want_dense)
        wav = wav.contiguous()
        B, N = wav.shape
        T = lib.dissc_hubert_frames(N)
        if T <= 0:
            raise ValueError(f"{N} samples are too short for one HuBERT frame (need >= 400)")
        ldT = (T + 3) // 4 * 4
        ns = None
        if n_samples is not None:
            ns = torch.as_tensor(n_samples)
            if ns.numel() != B:
                raise ValueError("n_samples must be [B]")
            if ns.device.type == "cpu" and (int(ns.min()) < 0 or int(ns.max()) > N):
                raise ValueError(f"n_samples must lie in [0, {N}] (row length of wav)")
            ns = ns.to(dev, torch.int32).contiguous()
        # frames per utterance (host arithmetic), uploaded BEFORE the forward's kernels are queued: an H2D copy from pageable memory
        # blocks the host until everything queued before it has run -- behind the encoder's launches it would block for the whole
        # encode and keep the caller from queueing its next stage (the Converter's predictors) under it
        if n_samples is None:
            frames = torch.full((B,), T, dtype=torch.int32)
        else:
            frames = torch.tensor([max(lib.dissc_hubert_frames(int(n)), 0) for n in torch.as_tensor(n_samples).tolist()],
                                  dtype=torch.int32)
        frames_dev = frames.to(dev)
        with torch.cuda.device(dev):
            need = lib.dissc_hubert_workspace_bytes(self._handle, B, N)
            if self._ws is None or self._ws.numel() < need:
                self._ws = None
                self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
            dense = torch.zeros(B, 768, ldT, dtype=torch.float32, device=dev) if want_dense else None
            units = torch.zeros(B, T, dtype=torch.int64, device=dev) if self.centers is not None else None
            check(lib.dissc_hubert_forward(self._handle, wav.data_ptr(), ns.data_ptr() if ns is not None else None,
                                           B, N, dense.data_ptr() if dense is not None else None,
                                           units.data_ptr() if units is not None else None, self._ws.data_ptr(),
                                           need, _lib.current_stream_ptr(dev)), "dissc_hubert_forward")
        out = {"frames": frames, "frames_dev": frames_dev}
        if units is not None:
            out["units"] = units
        if dense is not None:
            out["dense"] = dense[:, :, :T].transpose(1, 2)
        return out

    def _forward_chunked(self, wav, n_samples, want_dense):
        """Inputs longer than MAX_CHUNK: every chunk is encoded on its own and the frames are
        concatenated, as the reference's feature reader does (the frames next to a chunk boundary
        therefore differ from a single pass, on purpose).  A trailing piece shorter than one frame
        (< 400 samples) yields nothing."""
        B, N = wav.shape
        ns = torch.full((B,), N, dtype=torch.int64) if n_samples is None else torch.as_tensor(n_samples).cpu().long()
        parts = []
        for start in range(0, N, self.MAX_CHUNK):
            piece = wav[:, start:start + self.MAX_CHUNK]
            n_c = (ns - start).clamp(0, piece.shape[1])
            if piece.shape[1] < 400 or int(n_c.max()) < 400:
                continue
            parts.append(self.forward(piece.contiguous(), n_samples=n_c.to(torch.int32), want_dense=want_dense))
        frames = torch.stack([p["frames"] for p in parts]).sum(0).to(torch.int32)
        T = int(frames.max())
        out = {"frames": frames, "frames_dev": torch.stack([p["frames_dev"] for p in parts]).sum(0).to(torch.int32)}
        for key, shape, dt in (("units", (B, T), torch.int64), ("dense", (B, T, 768), torch.float32)):
            if key not in parts[0]:
                continue
            full = torch.zeros(shape, dtype=dt, device=self.device)
            for b in range(B):
                pos = 0
                for p in parts:
                    f = int(p["frames"][b])
                    full[b, pos:pos + f] = p[key][b, :f]
                    pos += f
            out[key] = full
        return out

    __call__ = forward


class SpeechEncoder:
    """textless-style front end: one waveform -> {'units','dense','durations','f0'}."""

    def __init__(self, dense_model, deduplicate=False):
        self.model = dense_model
        self.deduplicate = deduplicate

    @classmethod
    def from_files(cls, hubert_ckpt, kmeans_file, layer=6, deduplicate=False):
        sd = torch.load(hubert_ckpt, map_location="cpu", weights_only=False)
        return cls(HubertEncoder(sd, load_kmeans_centers(kmeans_file), n_layers=layer), deduplicate)

    @classmethod
    def by_name(cls, dense_model_name="hubert-base-ls960", quantizer_model_name="kmeans", vocab_size=100,
                deduplicate=False, checkpoint_dir=None):
        """textless downloads its checkpoints; offline we look them up in ``checkpoint_dir`` /
        $DISSC_CHECKPOINT_DIR: <dir>/<dense_model_name>.pt and <dir>/<quantizer>_<vocab>.{npy,bin,pt}"""
        d = checkpoint_dir or os.environ.get("DISSC_CHECKPOINT_DIR", "checkpoints/hubert")
        dense = os.path.join(d, dense_model_name + ".pt")
        km = next((os.path.join(d, f"{quantizer_model_name}_{vocab_size}{ext}") for ext in (".npy", ".bin", ".pt")
                   if os.path.exists(os.path.join(d, f"{quantizer_model_name}_{vocab_size}{ext}"))), None)
        if not os.path.exists(dense) or km is None:
            raise FileNotFoundError(f"need {dense} and {quantizer_model_name}_{vocab_size}.[npy|bin|pt] in {d} "
                                    "(no network access to download them)")
        return cls.from_files(dense, km, layer=6, deduplicate=deduplicate)

    def to(self, device):
        self.model.to(device)
        return self

    def __call__(self, waveform):
        out = self.model(waveform)
        units = out["units"][0]
        res = {"units": units, "dense": out["dense"][0], "durations": torch.ones_like(units),
               "f0": torch.zeros(units.shape[0], dtype=torch.float32, device=units.device)}
        if self.deduplicate:
            from .predictors import dedup
            vals, counts, n = dedup(units.unsqueeze(0))
            k = int(n[0])
            res["units"], res["durations"] = vals[0, :k], counts[0, :k].long()
        return res
