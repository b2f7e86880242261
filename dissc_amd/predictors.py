"""Drop-ins for the reference's predictors (reference model/len_predictor.py,
model/pitch_predictor.py) and for infer.py's per-sample logic, on the MI355X.

Same constructors, checkpoint formats (``best_model.pth`` state dicts, 53 / 34 / 78 keys) and
call signatures as the reference::

    len_model = LenPredictor(n_tokens=100, n_speakers=108).to('cuda:0'); len_model.eval()
    len_model.load_state_dict(torch.load('len/best_model.pth'))
    len_model.norm_mean, len_model.norm_std = torch.load('len/len_norm_stats.pth')
    lens = len_model(dd_seq, spk_id)                      # [B,L] f32
    f0 = pitch_model.infer_freq(out_seq, spk_id, norm)    # [B,T] f32

Beyond the reference every call takes ``lengths=`` for a ragged batch, and
``infer_samples`` runs dedup -> length -> carry-over -> expand -> pitch for a whole batch of
(utterance x target speaker) pairs with a single host synchronisation (the reference's
``len_carryover_correction`` alone syncs once per unit, infer.py:162-171).
"""
import ctypes

import torch

from . import _lib
from ._lib import lib, check

_BN_EPS = 1e-5


def _fold_bn(sd, bn):
    """eval-mode BatchNorm1d exactly as PyTorch's CPU kernel evaluates it:
    alpha = weight * (1/sqrt(var+eps)); beta = bias - mean*alpha; y = x*alpha + beta."""
    invstd = 1.0 / torch.sqrt(sd[bn + ".running_var"].float() + _BN_EPS)
    alpha = sd[bn + ".weight"].float() * invstd
    beta = sd[bn + ".bias"].float() - sd[bn + ".running_mean"].float() * alpha
    return alpha, beta


class _Predictor:
    _KIND = None
    _BN = {}  # conv name -> bn name

    def __init__(self):
        self.device = None
        self._sd = None
        self._handle = None
        self._ws = None
        self.training = False

    # nn.Module-like surface -------------------------------------------------------------
    def load_state_dict(self, sd, strict=True):
        need = set(self._expected_keys())
        got = set(sd.keys())
        missing = sorted(need - got)
        unexpected = sorted(got - need)
        if missing or (strict and unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}: "
                               f"missing {missing[:5]}, unexpected {unexpected[:5]}")
        self._sd = {k: v.detach().to("cpu") for k, v in sd.items()}
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
            raise _lib.DisscError(f"dissc_amd.{type(self).__name__} runs on an MI355X only")
        return self

    def _conv_names(self):
        raise NotImplementedError

    def _expected_keys(self):
        keys = ["token_emb.weight", "spk_emb.weight"]
        for c in self._conv_names():
            keys += [c + ".weight", c + ".bias"]
            if c in self._BN:
                b = self._BN[c]
                keys += [b + ".weight", b + ".bias", b + ".running_mean", b + ".running_var",
                         b + ".num_batches_tracked"]
        return keys

    def _tensors(self):
        t = {k: v.float() for k, v in self._sd.items()
             if k in ("token_emb.weight", "spk_emb.weight", "pe.pe")}
        for c in self._conv_names():
            t[c + ".weight"] = self._sd[c + ".weight"].float()
            t[c + ".bias"] = self._sd[c + ".bias"].float()
            if c in self._BN:
                t[c + ".bn_scale"], t[c + ".bn_shift"] = _fold_bn(self._sd, self._BN[c])
        return t

    def _ensure(self):
        if self._handle is not None:
            return
        if self._sd is None:
            raise RuntimeError("load_state_dict() first")
        if self.device is None:
            self.to("cuda:0")
        with torch.cuda.device(self.device):
            table, keep = _lib.make_tensor_table(self._tensors())
            h = ctypes.c_void_p()
            check(lib.dissc_pred_create(self._KIND, table, len(keep), ctypes.byref(h)), "dissc_pred_create")
            self._handle = h
        self._after_create()

    def _after_create(self):
        pass

    def _destroy(self):
        if self._handle is not None:
            lib.dissc_pred_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _workspace(self, B, L):
        need = lib.dissc_pred_workspace_bytes(self._handle, B, L)
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.device:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws, need

    @staticmethod
    def _check_ids(t, n_rows, what):
        """nn.Embedding raises IndexError on an out-of-range id (the reference's behaviour for a
        corrupt manifest / wrong speaker table); the kernels would clamp.  Checked for host tensors
        (ids that come from files); device-resident ids come from our own kernels and are trusted."""
        if t.device.type == "cpu" and t.numel():
            lo, hi = int(t.min()), int(t.max())
            if lo < 0 or hi >= n_rows:
                raise IndexError(f"{what} id out of range: [{lo}, {hi}] not within [0, {n_rows})")

    def _prep(self, seq, spk_id, lengths):
        dev = self.device
        seq = torch.as_tensor(seq)
        spk_id = torch.as_tensor(spk_id)
        if lengths is None:
            self._check_ids(seq, self._sd["token_emb.weight"].shape[0], "unit")
        self._check_ids(spk_id, self._sd["spk_emb.weight"].shape[0], "speaker")
        seq = seq.to(dev, torch.int64)
        if seq.dim() == 1:
            seq = seq.unsqueeze(0)
        seq = seq.contiguous()
        B = seq.shape[0]
        spk = torch.as_tensor(spk_id).to(dev, torch.int64).reshape(-1).contiguous()
        if spk.numel() != B:
            raise ValueError("spk_id must be [B,1]")
        if lengths is not None:
            lengths = torch.as_tensor(lengths).to(dev, torch.int32).contiguous()
        return seq, spk, lengths


class LenPredictor(_Predictor):
    """reference model/len_predictor.py:5-52"""
    _KIND = 0
    _BN = {"cnn1": "bn1", **{f"cnn1{i}": f"bn1{i}" for i in range(1, 7)}}

    def __init__(self, n_tokens=100, n_speakers=99, emb_size=32, masking_rate=0.2,
                 norm_mean=torch.tensor(0), norm_std=torch.tensor(1)):
        super().__init__()
        self.n_tokens, self.n_speakers = n_tokens, n_speakers
        self._norm = None
        self.norm_mean, self.norm_std = norm_mean, norm_std

    def _conv_names(self):
        return ["cnn1"] + [f"cnn1{i}" for i in range(1, 7)] + ["cnn2"]

    def _sync_norm(self):
        cur = (float(self.norm_mean), float(self.norm_std))
        if cur != self._norm:
            check(lib.dissc_len_set_norm(self._handle, cur[0], cur[1]), "dissc_len_set_norm")
            self._norm = cur

    def _after_create(self):
        self._norm = None

    def forward(self, seq, spk_id, lengths=None):
        self._ensure()
        seq, spk, lengths = self._prep(seq, spk_id, lengths)
        B, L = seq.shape
        ldo = (L + 3) // 4 * 4
        with torch.cuda.device(self.device):
            self._sync_norm()
            out = torch.zeros(B, ldo, dtype=torch.float32, device=self.device)
            ws, need = self._workspace(B, L)
            check(lib.dissc_len_forward(self._handle, seq.data_ptr(), spk.data_ptr(),
                                        lengths.data_ptr() if lengths is not None else None, B, L,
                                        out.data_ptr(), ldo, ws.data_ptr(), need,
                                        _lib.current_stream_ptr(self.device)), "dissc_len_forward")
        return out[:, :L]

    __call__ = forward


class _PitchBase(_Predictor):
    def __init__(self, n_tokens=100, n_speakers=199, emb_size=32, masking_rate=0.4,
                 id2pitch_mean=None, id2pitch_std=None):
        super().__init__()
        self.n_tokens, self.n_speakers = n_tokens, n_speakers
        self.id2pitch_mean, self.id2pitch_std = id2pitch_mean, id2pitch_std
        self._stats_dev = None

    def infer_freq(self, seq, spk_id, norm=False, lengths=None):
        self._ensure()
        seq, spk, lengths = self._prep(seq, spk_id, lengths)
        B, T = seq.shape
        mean = std = None
        if not norm:
            if self.id2pitch_mean is None:
                raise ValueError("id2pitch_mean/std are needed for un-normalised pitch")
            self._check_ids(torch.as_tensor(spk_id), len(self.id2pitch_mean), "speaker (id2pitch_mean)")
            if self._stats_dev is None:
                self._stats_dev = (torch.as_tensor(self.id2pitch_mean).to(self.device, torch.float32).contiguous(),
                                   torch.as_tensor(self.id2pitch_std).to(self.device, torch.float32).contiguous())
            mean, std = self._stats_dev
        ldo = (T + 3) // 4 * 4
        with torch.cuda.device(self.device):
            out = torch.zeros(B, ldo, dtype=torch.float32, device=self.device)
            ws, need = self._workspace(B, T)
            check(lib.dissc_pitch_forward(self._handle, seq.data_ptr(), spk.data_ptr(),
                                          lengths.data_ptr() if lengths is not None else None, B, T,
                                          1 if norm else 0, mean.data_ptr() if mean is not None else None,
                                          std.data_ptr() if std is not None else None,
                                          int(mean.numel()) if mean is not None else 0, out.data_ptr(), ldo,
                                          ws.data_ptr(), need, _lib.current_stream_ptr(self.device)),
                  "dissc_pitch_forward")
        return out[:, :T]


class PitchPredictor(_PitchBase):
    """"new" variant: positional encoding on the speaker embedding, BatchNorm after cnn2 only
    (reference model/pitch_predictor.py:41-104)."""
    _KIND = 1
    _BN = {"cnn2": "bn2"}

    def _conv_names(self):
        return ["cnn1"] + [f"cnn1{i}" for i in range(1, 8)] + ["cnn2", "cnn_class1", "cnn_class2",
                                                                "cnn_reg1", "cnn_reg2"]

    def _expected_keys(self):
        return super()._expected_keys() + ["pe.pe"]


class PitchPredictorBase(_PitchBase):
    """"base" variant: BatchNorm after every conv but cnn2, no positional encoding
    (reference model/pitch_predictor.py:106-176)."""
    _KIND = 2
    _BN = {"cnn1": "bn1", **{f"cnn1{i}": f"bn1{i}" for i in range(1, 8)},
           "cnn_class1": "bn_c1", "cnn_reg1": "bn_r1"}

    def _conv_names(self):
        return ["cnn1"] + [f"cnn1{i}" for i in range(1, 8)] + ["cnn2", "cnn_class1", "cnn_class2",
                                                                "cnn_reg1", "cnn_reg2"]


# ---------------------------------------------------------------------------------------------
# infer.py's sample logic, batched (reference infer.py:24-45)
# ---------------------------------------------------------------------------------------------
def dedup(units, lengths=None):
    """units i64 [B,T] (cuda) -> (vals i64 [B,T], counts i32 [B,T], n i32 [B])"""
    units = units.contiguous()
    B, T = units.shape
    dev = units.device
    vals = torch.zeros(B, T, dtype=torch.int64, device=dev)
    counts = torch.zeros(B, T, dtype=torch.int32, device=dev)
    n = torch.zeros(B, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.dissc_dedup(units.data_ptr(), lengths.data_ptr() if lengths is not None else None, B, T,
                              vals.data_ptr(), counts.data_ptr(), n.data_ptr(), _lib.current_stream_ptr(dev)),
              "dissc_dedup")
    return vals, counts, n


def len_carryover_correction(lens, n=None):
    """lens f32 [B,L] (cuda, row stride = L) -> (lens_int i32 [B,L], totals i32 [B])"""
    lens = lens.contiguous()
    B, L = lens.shape
    dev = lens.device
    if n is None:
        n = torch.full((B,), L, dtype=torch.int32, device=dev)
    out = torch.zeros(B, L, dtype=torch.int32, device=dev)
    totals = torch.zeros(B, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.dissc_len_carryover(lens.data_ptr(), n.data_ptr(), B, L, out.data_ptr(), totals.data_ptr(),
                                      _lib.current_stream_ptr(dev)), "dissc_len_carryover")
    return out, totals


def expand(vals, lens_int, n, t_out):
    """repeat_interleave per row -> i64 [B,t_out]"""
    B, L = vals.shape
    dev = vals.device
    out = torch.zeros(B, max(int(t_out), 1), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        check(lib.dissc_expand(vals.data_ptr(), lens_int.data_ptr(), n.data_ptr(), B, L, out.data_ptr(),
                               out.shape[1], _lib.current_stream_ptr(dev)), "dissc_expand")
    return out


def infer_batch(units, lengths, spk, len_model=None, pitch_model=None, norm_pitch=False):
    """Device-resident ``_infer_sample`` for a ragged batch (reference infer.py:24-45): everything stays
    in HBM, ONE host synchronisation (the expanded lengths, needed to size the outputs).

    units i64 [B,T] (cuda; pad tokens already removed), lengths i32 [B] (cuda), spk i64 [B] or [B,1]
    (cuda or host).  Returns a dict: ``units`` i64 [B,T'] / ``lengths`` i32 [B] (cuda) = the frame-rate
    unit sequence after the rhythm model (the input when there is none), ``totals`` = the same lengths
    on the host, ``f0`` f32 [B,T'] (cuda) or None, ``lens_int`` i32 [B,T] / ``n`` i32 [B] (cuda) = frames
    per dedup'd unit or None."""
    dev = units.device
    B = units.shape[0]
    spk = torch.as_tensor(spk).reshape(B, 1)
    if len_model is not None:
        vals, _, n = dedup(units, lengths)
        lens = len_model(vals, spk, lengths=n)            # [B,T] (only the first n[b] are valid)
        lens_int, totals = len_carryover_correction(lens.contiguous(), n)
        tot = totals.cpu()                                # the one host sync of the batch
        t_out = int(tot.max()) if B else 0
        out_seq = expand(vals, lens_int, n, t_out)
        out_len = totals
    else:
        out_seq, out_len, tot = units, lengths, lengths.cpu()
        lens_int = n = None
    f0 = None
    if pitch_model is not None and B and int(tot.max()) > 0:
        f0 = pitch_model.infer_freq(out_seq, spk, norm_pitch, lengths=out_len)
    return {"units": out_seq, "lengths": out_len, "totals": tot, "f0": f0, "lens_int": lens_int, "n": n}


def infer_samples(unit_seqs, spk_ids, len_model=None, pitch_model=None, norm_pitch=False, n_tokens=100,
                  device="cuda:0"):
    """Batched ``_infer_sample`` for the --pred_len/--pred_pitch modes (list front end of infer_batch).

    unit_seqs: list of 1-D int sequences (pad token n_tokens is dropped like infer.py:25);
    spk_ids: list of target speaker ids.  Returns a list of
    (units list[int], f0 list[float]|None, lens list[int]|None) -- lens = corrected frames per
    dedup'd unit when a length model is given."""
    dev = torch.device(device)
    seqs = [torch.as_tensor(s).long().reshape(-1) for s in unit_seqs]
    seqs = [s[s != n_tokens] for s in seqs]
    B = len(seqs)
    if B == 0:
        return []
    for m in (len_model, pitch_model):
        if m is not None and m._sd is not None:
            for s in seqs:
                m._check_ids(s, m._sd["token_emb.weight"].shape[0], "unit")
            m._check_ids(torch.as_tensor(spk_ids), m._sd["spk_emb.weight"].shape[0], "speaker")
    lengths = torch.tensor([len(s) for s in seqs], dtype=torch.int32)
    T = max(int(lengths.max()), 1)
    units = torch.full((B, T), 0, dtype=torch.int64)
    for i, s in enumerate(seqs):
        units[i, :len(s)] = s
    spk = torch.as_tensor(spk_ids, dtype=torch.int64).reshape(B, 1)
    r = infer_batch(units.to(dev), lengths.to(dev), spk, len_model, pitch_model, norm_pitch)
    tot = r["totals"]
    out_seq_c = r["units"].cpu()
    f0 = r["f0"].cpu() if r["f0"] is not None else None
    lens_c = r["lens_int"].cpu() if r["lens_int"] is not None else None
    n_c = r["n"].cpu() if r["n"] is not None else None
    res = []
    for i in range(B):
        k = int(tot[i])
        res.append((out_seq_c[i, :k].tolist(), f0[i, :k].tolist() if f0 is not None else None,
                    lens_c[i, :int(n_c[i])].tolist() if lens_c is not None else None))
    return res
