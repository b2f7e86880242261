"""YAAPT F0 tracking on the MI355X: what ``data/encode.py`` gets as ``'f0'`` from textless' SpeechEncoder and
what ``get_yaapt_f0`` computes in the reference (reference sr/dataset.py:27-43, data/encode.py:32-38).

    trk = YaaptTracker(device='cuda:0')
    f0s = trk(list_of_waveforms)            # one float32 array per utterance, a value per 5 ms, 0 = unvoiced
    per_unit = f0_per_unit(f0s[0], n_units) # 20 ms units (textless alignment)

The heavy stages run in libdissc_hip.so (csrc/yaapt.hip through the C ABI ``dissc_yaapt_*``): band-pass FIR of
the signal and its square, frame spectra as a DFT-GEMM on the matrix cores, NLFER band energy, spectral harmonics
correlation + its peak candidates, NCCF + its peak candidates, for a whole ragged batch per launch.  The short,
inherently sequential stages (median smoothing, the two dynamic-programming passes, pchip interpolation, merging
of the candidate sets) are host logic below, vectorised over frames.

PARITY UNPINNED: the reference's tracker is amfm_decompy.pYAAPT, an un-vendored third party that is not
available offline; the algorithm (Zahorian & Hu 2008, with amfm_decompy's parameter table and the reference's
four overrides) is restated in oracle/yaapt_ref.py and this module is tested against that restatement and
against known-F0 signals (tests/test_yaapt.py, tests/test_gpu_yaapt.py).
"""
import ctypes

import numpy as np
import torch
from scipy import interpolate as _interp
from scipy.signal import firwin, lfilter, medfilt

from . import _lib
from ._lib import check, lib

# amfm_decompy defaults [3P-unverified] with the reference's overrides (frame_length 20, frame_space 5,
# nccf_thresh1 0.25, tda_frame_length 25: reference sr/dataset.py:35-36)
DEFAULTS = dict(frame_length=20.0, tda_frame_length=25.0, frame_space=5.0, f0_min=60.0, f0_max=400.0, fft_length=8192,
                bp_forder=150, bp_low=50.0, bp_high=1500.0, nlfer_thresh1=0.75, nlfer_thresh2=0.1, shc_numharms=3,
                shc_window=40.0, shc_maxpeaks=4, shc_pwidth=50.0, shc_thresh1=5.0, shc_thresh2=1.25, f0_double=150.0,
                f0_half=150.0, dp5_k1=11.0, nccf_thresh1=0.25, nccf_thresh2=0.9, nccf_maxcands=3, nccf_pwidth=5,
                merit_boost=0.20, merit_pivot=0.99, merit_extra=0.4, median_value=7, dp_w1=0.15, dp_w2=0.5, dp_w3=0.1,
                dp_w4=0.9, spec_pitch_min_std=0.05)


class _Cfg(ctypes.Structure):
    _fields_ = [("fs", ctypes.c_int32), ("frame_len", ctypes.c_int32), ("frame_hop", ctypes.c_int32),
                ("tda_len", ctypes.c_int32), ("nfft", ctypes.c_int32), ("f0_min", ctypes.c_float),
                ("f0_max", ctypes.c_float), ("shc_numharms", ctypes.c_int32), ("shc_window_hz", ctypes.c_float),
                ("shc_pwidth_hz", ctypes.c_float), ("shc_thresh1", ctypes.c_float), ("shc_thresh2", ctypes.c_float),
                ("f0_double", ctypes.c_float), ("f0_half", ctypes.c_float), ("nccf_thresh1", ctypes.c_float),
                ("nccf_thresh2", ctypes.c_float), ("nccf_pwidth", ctypes.c_int32)]


class _TrackCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in ("nlfer_thresh1", "nlfer_thresh2", "dp5_k1", "merit_boost", "merit_pivot",
                                                "merit_extra", "dp_w1", "dp_w2", "dp_w3", "dp_w4", "spec_pitch_min_std",
                                                "f0_min", "f0_max")] + \
               [(n, ctypes.c_int32) for n in ("median_value", "nccf_pwidth", "fs", "reserved")]


def _bind():
    vp, i32 = ctypes.c_void_p, ctypes.c_int
    lib.dissc_yaapt_track_workspace_bytes.argtypes = [i32, i32]
    lib.dissc_yaapt_track_workspace_bytes.restype = ctypes.c_size_t
    lib.dissc_yaapt_spec_track.argtypes = [ctypes.POINTER(_TrackCfg)] + [vp] * 5 + [i32, i32] + [vp] * 7 + \
                                          [ctypes.c_size_t, vp]
    lib.dissc_yaapt_final_track.argtypes = [ctypes.POINTER(_TrackCfg)] + [vp] * 9 + [i32, i32, vp, vp, ctypes.c_size_t, vp]
    lib.dissc_yaapt_create.argtypes = [vp, i32, ctypes.POINTER(_Cfg), ctypes.POINTER(vp)]
    lib.dissc_yaapt_destroy.argtypes = [vp]
    lib.dissc_yaapt_destroy.restype = None
    for fn in (lib.dissc_yaapt_frames, lib.dissc_yaapt_tda_frames):
        fn.argtypes = [vp, i32]
    lib.dissc_yaapt_shc_bins.argtypes = [vp]
    lib.dissc_yaapt_workspace_bytes.argtypes = [vp, i32, i32]
    lib.dissc_yaapt_workspace_bytes.restype = ctypes.c_size_t
    lib.dissc_yaapt_spectral.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    lib.dissc_yaapt_nccf.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, ctypes.c_size_t, vp]


_bind()


# ---------------------------------------------------------------------------------------------------------
# host stages (numpy, one utterance at a time, vectorised over frames)
# ---------------------------------------------------------------------------------------------------------
def _viterbi(local, trans):
    """local [K,F], trans [F-1,K(to),K(from)] -> best state per frame (first minimum on ties)"""
    K, F = local.shape
    cum = local[:, 0].copy()
    back = np.zeros((F, K), dtype=np.int64)
    rows = np.arange(K)
    for i in range(1, F):
        tot = trans[i - 1] + cum[None, :]
        back[i] = np.argmin(tot, axis=1)
        cum = tot[rows, back[i]] + local[:, i]
    path = np.empty(F, dtype=np.int64)
    path[-1] = int(np.argmin(cum))
    for i in range(F - 1, 0, -1):
        path[i - 1] = back[i, path[i]]
    return path


def spectral_track(cand_pitch, cand_merit, p):
    """SHC candidates [4,F] (pitch 0 = no candidate) -> smooth spectral F0 track [F] and its spread."""
    F = cand_pitch.shape[1]
    voiced = cand_pitch[0] > 0
    nv = int(voiced.sum())
    if nv == 0:
        return np.full(F, 150.0), 150.0 * p["spec_pitch_min_std"]
    vp, vm = cand_pitch[:, voiced].astype(np.float64), cand_merit[:, voiced].astype(np.float64)
    avg, std = vp[0].mean(), vp[0].std()
    cols = np.arange(nv)
    pick = np.argmin(np.abs(vp - 0.8 * avg) * (3.0 - vm), axis=0)
    k5 = max(1, int(p["median_value"]) - 2)
    smooth, keep = medfilt(vp[pick, cols], k5), vm[pick, cols].copy()
    vp[pick, cols], vm[pick, cols] = smooth, keep
    if nv > 2:
        trans = p["dp5_k1"] * std / avg * np.abs(vp.T[1:, :, None] - vp.T[:-1, None, :]) / p["f0_min"]
        track = medfilt(vp[_viterbi(1.0 - vm, trans), cols], k5)
    else:
        track = np.full(nv, 150.0)
    t_avg = track.mean()
    t_std = max(track.std(), t_avg * p["spec_pitch_min_std"])
    spec = np.zeros(F)
    spec[voiced] = track
    if spec[0] < t_avg / 2:
        spec[0] = t_avg
    if spec[-1] < t_avg / 2:
        spec[-1] = t_avg
    nz = np.nonzero(spec)[0]
    spec = _interp.pchip(nz, spec[nz])(np.arange(F)) if len(nz) > 1 else np.full(F, t_avg)
    spec = lfilter(np.ones(3) / 3.0, 1.0, spec)
    if F > 3:
        spec[0], spec[1] = spec[2], spec[3]
    return spec, t_std


def lag_ranges(spec, std, fs, p):
    lo = np.maximum(spec - 2.0 * std, p["f0_min"])
    hi = np.minimum(spec + 2.0 * std, p["f0_max"])
    half = int(p["nccf_pwidth"]) // 2
    return np.fix(fs / hi).astype(np.int32) - half, np.fix(fs / lo).astype(np.int32) + half


def merge_candidates(tp1, tm1, tp2, tm2, spec, std, energy, vuv, p):
    """NCCF candidates of both signals [3,F] each -> the candidate table of the final DP [8,F]"""
    F = tp1.shape[1]
    thresh = 5.0 * std
    pitch = np.vstack([tp1, tp2]).astype(np.float64)
    diff = np.abs(pitch - spec[None, :F])
    merit = (1.0 + p["merit_boost"]) * np.vstack([tm1, tm2]) * np.where(diff < thresh, 1.0 - diff / thresh, 0.0)
    order = np.argsort(-merit, axis=0, kind="stable")
    cols = np.arange(F)[None, :]
    pitch, merit = pitch[order, cols], merit[order, cols]
    K = pitch.shape[0]
    en, vu, sp = energy[:F], vuv[:F], spec[:F]
    best = medfilt(pitch[0], int(p["median_value"])) * vu
    dead = en <= p["nlfer_thresh2"]
    has = (pitch[0] > 0) & ~dead
    none = ~has & ~dead
    m0 = merit[0].copy()
    # voiced candidate present: the last row becomes the unvoiced option, empty middle rows lose their merit
    mid = slice(1, K - 1)
    merit[mid] = np.where(has[None, :] & (pitch[mid] == 0), 0.0, merit[mid])
    pitch[K - 1] = np.where(has, 0.0, pitch[K - 1])
    merit[K - 1] = np.where(has, 1.0 - m0, merit[K - 1])
    # nothing from the NCCF: the spectral estimate with a merit from the energy, the rest unvoiced
    fall = np.minimum(1.0, en / 2.0)
    pitch[0] = np.where(none, sp, pitch[0])
    merit[0] = np.where(none, fall, merit[0])
    pitch[1:] = np.where(none[None, :], 0.0, pitch[1:])
    merit[1:] = np.where(none[None, :], (1.0 - fall)[None, :], merit[1:])
    # definitely unvoiced frames
    pitch = np.where(dead[None, :], 0.0, pitch)
    merit = np.where(dead[None, :], p["merit_pivot"], merit)
    extra_p = np.vstack([best, sp * vu])
    extra_m = np.vstack([np.where(best > 0, p["merit_extra"], 0.0), np.where(vu, p["merit_extra"], 0.0)])
    return np.vstack([pitch[:K - 1], extra_p, pitch[K - 1:]]), np.vstack([merit[:K - 1], extra_m, merit[K - 1:]])


def final_track(pitch, merit, energy, p):
    """the final dynamic-programming pass over the candidate table -> F0 per frame (0 = unvoiced)"""
    K, F = pitch.shape
    best = pitch[0]
    mean_pitch = best[best > 0].mean() if np.any(best > 0) else 150.0
    cur, prv = pitch.T[1:, :, None], pitch.T[:-1, None, :]
    both, neither = (cur > 0) & (prv > 0), (cur == 0) & (prv == 0)
    benefit = np.minimum(1.0, np.abs(energy[:F - 1] - energy[1:F]))
    trans = np.where(both, p["dp_w1"] * np.abs(cur - prv) / mean_pitch,
                     np.where(neither, p["dp_w3"], (p["dp_w2"] * (1.0 - benefit))[:, None, None]))
    path = _viterbi(p["dp_w4"] * (1.0 - merit), trans)
    return pitch[path, np.arange(F)]


def f0_per_unit(f0_frames, n_units, ratio=4):
    """textless' alignment of the 5 ms track to 20 ms units (align_f0_to_durations [3P-unverified]): the track is cut
    to ratio * n_units frames or, when shorter, extended with its LAST value (not with zeros: the last unit keeps
    the voicing of the track's end); unit i = frames [4i, 4i+4), mean of the voiced values, 0.0 when there is none"""
    f = np.zeros(n_units * ratio, dtype=np.float64)
    m = min(len(f0_frames), len(f))
    f[:m] = np.asarray(f0_frames[:m], dtype=np.float64)
    if 0 < m < len(f):
        f[m:] = f[m - 1]
    f = f.reshape(n_units, ratio)
    cnt = (f != 0).sum(1)
    return np.where(cnt > 0, f.sum(1) / np.maximum(cnt, 1), 0.0)


# ---------------------------------------------------------------------------------------------------------
class YaaptTracker:
    def __init__(self, device="cuda:0", fs=16000, **overrides):
        self.p = dict(DEFAULTS, **overrides)
        self.fs = int(fs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.DisscError("dissc_amd.f0.YaaptTracker runs on an MI355X only")
        p = self.p
        self.flen = int(p["frame_length"] * fs / 1000)
        self.hop = int(p["frame_space"] * fs / 1000)
        self.tda = int(p["tda_frame_length"] * fs / 1000)
        fir = firwin(int(p["bp_forder"]) + 1, [p["bp_low"] / (fs / 2), p["bp_high"] / (fs / 2)], pass_zero=False)
        self._fir = np.ascontiguousarray(fir, dtype=np.float32)
        cfg = _Cfg(self.fs, self.flen, self.hop, self.tda, int(p["fft_length"]), p["f0_min"], p["f0_max"],
                   int(p["shc_numharms"]), p["shc_window"], p["shc_pwidth"], p["shc_thresh1"], p["shc_thresh2"],
                   p["f0_double"], p["f0_half"], p["nccf_thresh1"], p["nccf_thresh2"], int(p["nccf_pwidth"]))
        if int(p["shc_maxpeaks"]) != 4 or int(p["nccf_maxcands"]) != 3:
            raise ValueError("the kernels are built for shc_maxpeaks = 4 and nccf_maxcands = 3")
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.dissc_yaapt_create(self._fir.ctypes.data, len(self._fir), ctypes.byref(cfg), ctypes.byref(h)),
                  "dissc_yaapt_create")
        self._h = h
        self._ws = None
        self._tws = None
        self._tcfg = _TrackCfg(p["nlfer_thresh1"], p["nlfer_thresh2"], p["dp5_k1"], p["merit_boost"], p["merit_pivot"],
                               p["merit_extra"], p["dp_w1"], p["dp_w2"], p["dp_w3"], p["dp_w4"], p["spec_pitch_min_std"],
                               p["f0_min"], p["f0_max"], int(p["median_value"]), int(p["nccf_pwidth"]), self.fs, 0)

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None:
                lib.dissc_yaapt_destroy(self._h)
        except Exception:
            pass

    def _workspace(self, B, N):
        need = lib.dissc_yaapt_workspace_bytes(self._h, B, N)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws, need

    # -- device stages ---------------------------------------------------------------------------------
    def spectral(self, wav, n_samples, want_shc=False):
        """wav f32 [B,N] (padded like the reference), n_samples i32 [B] -> dict of device tensors"""
        dev = self.device
        wav = torch.as_tensor(wav).to(dev, torch.float32).contiguous()
        B, N = wav.shape
        ns = torch.as_tensor(n_samples).to(dev, torch.int32).contiguous()
        F = lib.dissc_yaapt_frames(self._h, N)
        if F <= 0:
            raise ValueError(f"{N} samples are too short for one {self.flen}-sample frame")
        out = {"filt": torch.empty(B, N, device=dev), "nlfilt": torch.empty(B, N, device=dev),
               "energy": torch.empty(B, F, device=dev), "cand_pitch": torch.empty(B, F, 4, device=dev),
               "cand_merit": torch.empty(B, F, 4, device=dev)}
        shc = torch.empty(B, F, lib.dissc_yaapt_shc_bins(self._h), device=dev) if want_shc else None
        with torch.cuda.device(dev):
            ws, need = self._workspace(B, N)
            check(lib.dissc_yaapt_spectral(self._h, wav.data_ptr(), ns.data_ptr(), B, N, out["filt"].data_ptr(),
                                           out["nlfilt"].data_ptr(), out["energy"].data_ptr(),
                                           out["cand_pitch"].data_ptr(), out["cand_merit"].data_ptr(),
                                           shc.data_ptr() if shc is not None else None, ws.data_ptr(), need,
                                           _lib.current_stream_ptr(dev)), "dissc_yaapt_spectral")
        if shc is not None:
            out["shc"] = shc
        out["n_samples"], out["F"] = ns, F
        return out

    def nccf(self, sig, n_samples, lag_min, lag_max, want_phi=False):
        """sig f32 [B,N] (device), lag_min/lag_max int32 [B,F] -> (pitch [B,F,3], merit [B,F,3][, phi [B,F,tda]])"""
        dev = self.device
        B, N = sig.shape
        lmin = torch.as_tensor(lag_min).to(dev, torch.int32).contiguous()
        lmax = torch.as_tensor(lag_max).to(dev, torch.int32).contiguous()
        F = lmin.shape[1]
        pitch, merit = torch.empty(B, F, 3, device=dev), torch.empty(B, F, 3, device=dev)
        phi = torch.empty(B, F, self.tda, device=dev) if want_phi else None
        with torch.cuda.device(dev):
            ws, need = self._workspace(B, N)
            check(lib.dissc_yaapt_nccf(self._h, sig.data_ptr(), n_samples.data_ptr(), lmin.data_ptr(), lmax.data_ptr(),
                                       B, N, F, pitch.data_ptr(), merit.data_ptr(),
                                       phi.data_ptr() if phi is not None else None, ws.data_ptr(), need,
                                       _lib.current_stream_ptr(dev)), "dissc_yaapt_nccf")
        return (pitch, merit, phi) if want_phi else (pitch, merit)

    def spec_track(self, s, n_frames, n_tda):
        """device spec_track: the outputs of `spectral` + frame counts (int32 [B]) -> dict of device tensors
        (en_norm f64 [B,F], vuv u8 [B,F], spec f64 [B,F], spec_std f64 [B], lag_min / lag_max i32 [B,F])"""
        dev = self.device
        B, F = s["energy"].shape
        nf = torch.as_tensor(n_frames).to(dev, torch.int32).contiguous()
        nt = torch.as_tensor(n_tda).to(dev, torch.int32).contiguous()
        out = {"en_norm": torch.empty(B, F, dtype=torch.float64, device=dev),
               "vuv": torch.empty(B, F, dtype=torch.uint8, device=dev),
               "spec": torch.empty(B, F, dtype=torch.float64, device=dev),
               "spec_std": torch.empty(B, dtype=torch.float64, device=dev),
               "lag_min": torch.empty(B, F, dtype=torch.int32, device=dev),
               "lag_max": torch.empty(B, F, dtype=torch.int32, device=dev), "n_frames": nf, "n_tda": nt}
        with torch.cuda.device(dev):
            ws, need = self._track_workspace(B, F)
            check(lib.dissc_yaapt_spec_track(ctypes.byref(self._tcfg), s["energy"].data_ptr(), s["cand_pitch"].data_ptr(),
                                             s["cand_merit"].data_ptr(), nf.data_ptr(), nt.data_ptr(), B, F,
                                             out["en_norm"].data_ptr(), out["vuv"].data_ptr(), out["spec"].data_ptr(),
                                             out["spec_std"].data_ptr(), out["lag_min"].data_ptr(),
                                             out["lag_max"].data_ptr(), ws.data_ptr(), need,
                                             _lib.current_stream_ptr(dev)), "dissc_yaapt_spec_track")
        return out

    def final_track_device(self, st, c1, c2):
        """device merge + final dynamic programming: st = spec_track's dict, c1 / c2 = (pitch, merit) of the two NCCF
        passes -> f0 f32 [B,F] (device)"""
        dev = self.device
        B, F = st["spec"].shape
        f0 = torch.empty(B, F, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            ws, need = self._track_workspace(B, F)
            check(lib.dissc_yaapt_final_track(ctypes.byref(self._tcfg), c1[0].data_ptr(), c1[1].data_ptr(),
                                              c2[0].data_ptr(), c2[1].data_ptr(), st["en_norm"].data_ptr(),
                                              st["vuv"].data_ptr(), st["spec"].data_ptr(), st["spec_std"].data_ptr(),
                                              st["n_tda"].data_ptr(), B, F, f0.data_ptr(), ws.data_ptr(), need,
                                              _lib.current_stream_ptr(dev)), "dissc_yaapt_final_track")
        return f0

    def _track_workspace(self, B, F):
        need = lib.dissc_yaapt_track_workspace_bytes(B, F)
        if self._tws is None or self._tws.numel() < need:
            self._tws = None
            self._tws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._tws, need

    # -- the tracker -------------------------------------------------------------------------------------
    def __call__(self, waveforms, host_dp=False):
        """list of 1-D float waveforms @fs -> list of float32 F0 tracks (one value per frame_space, 0 = unvoiced),
        each of len(arange(frame/2, n + frame - frame/2, hop)) values like pYAAPT's samp_values on the padded signal.
        Everything runs on the device (one D2H copy of the tracks at the end); host_dp=True runs the sequential
        stages with the numpy functions above instead (kept as the readable form and for the tests)."""
        pad = self.flen // 2  # reference sr/dataset.py:29,33: 10 ms of zeros at both ends
        B = len(waveforms)
        if B == 0:
            return []
        lens = [len(w) + 2 * pad for w in waveforms]
        N = (max(lens) + 3) // 4 * 4
        # page-locked staging, allocated once and reused (page-locking 20 MB costs tens of ms; the H2D copy inside
        # `spectral` is synchronous with respect to the host, so the next call may overwrite it)
        if getattr(self, "_pin", None) is None or self._pin.numel() < B * N:
            self._pin = None
            self._pin = torch.empty(B * N, dtype=torch.float32, pin_memory=True)
        wav = self._pin[:B * N].view(B, N)
        for i, w in enumerate(waveforms):
            wav[i, :pad] = 0.0
            wav[i, pad:pad + len(w)] = torch.as_tensor(np.asarray(w, dtype=np.float32))
            wav[i, pad + len(w):] = 0.0
        s = self.spectral(wav, torch.tensor(lens, dtype=torch.int32))
        nfr = [lib.dissc_yaapt_frames(self._h, n) for n in lens]
        ntd = [min(lib.dissc_yaapt_tda_frames(self._h, n), f) for n, f in zip(lens, nfr)]
        if host_dp:
            return self._host_stages(s, nfr, ntd)
        st = self.spec_track(s, nfr, ntd)
        c1 = self.nccf(s["filt"], s["n_samples"], st["lag_min"], st["lag_max"])
        c2 = self.nccf(s["nlfilt"], s["n_samples"], st["lag_min"], st["lag_max"])
        f0 = self.final_track_device(st, c1, c2).cpu().numpy()
        return [f0[b, :nfr[b]].copy() for b in range(B)]

    def _host_stages(self, s, nfr, ntd):
        p = self.p
        B, F = len(nfr), s["F"]
        energy = s["energy"].cpu().numpy().astype(np.float64)
        cp = s["cand_pitch"].cpu().numpy()
        cm = s["cand_merit"].cpu().numpy()
        specs, stds, ens, vuvs = [], [], [], []
        lmin = np.ones((B, F), dtype=np.int32)
        lmax = np.full((B, F), 2, dtype=np.int32)
        for b in range(B):
            f = nfr[b]
            en = energy[b, :f]
            mean = en.mean() if f else 1.0
            en = en / mean if mean > 0 else en
            vuv = en > p["nlfer_thresh1"]
            pit = np.where(vuv[None, :], cp[b, :f].T, 0.0)
            mer = np.where(vuv[None, :], cm[b, :f].T, 1.0)
            spec, std = spectral_track(pit, mer, p) if f else (np.zeros(0), 1.0)
            lo, hi = lag_ranges(spec[:ntd[b]], std, self.fs, p)
            lmin[b, :ntd[b]], lmax[b, :ntd[b]] = lo, hi
            specs.append(spec), stds.append(std), ens.append(en), vuvs.append(vuv)
        tp1, tm1 = (t.cpu().numpy() for t in self.nccf(s["filt"], s["n_samples"], lmin, lmax))
        tp2, tm2 = (t.cpu().numpy() for t in self.nccf(s["nlfilt"], s["n_samples"], lmin, lmax))
        out = []
        for b in range(B):
            f, t = nfr[b], ntd[b]
            f0 = np.zeros(f, dtype=np.float32)
            if t > 0:
                rp, rm = merge_candidates(tp1[b, :t].T, tm1[b, :t].T, tp2[b, :t].T, tm2[b, :t].T, specs[b], stds[b],
                                          ens[b], vuvs[b], p)
                f0[:t] = final_track(rp, rm, ens[b], p)
            out.append(f0)
        return out
