"""Audio preparation before the path: what reference data/preprocess.py:19-36 does with
soundfile + resampy + librosa -- read, resample to 16 kHz, optional trim, optional zero padding to a multiple of
1280 samples, write.  The resampler runs on the MI355X (csrc/pipeline_glue.hip ``dissc_resample``, fp64 like
resampy); the filter table, the silence trimming (a few hundred frame energies) and the padding are host numpy.

PARITY UNPINNED for the resampler and the trim rule (resampy / librosa are un-vendored third parties that are
not available offline; restated in oracle/preprocess_ref.py); the pad rule is the reference's own numpy.
"""
import numpy as np
import torch
from scipy.io import wavfile
from scipy.signal.windows import kaiser

from ._lib import check, current_stream_ptr, lib

# resampy's shipped filters [3P-unverified]: zero crossings, table bits, roll-off, Kaiser beta
_FILTERS = {"kaiser_best": (64, 9, 0.9475937167399596, 14.769656459379492),
            "kaiser_fast": (16, 9, 0.85, 8.555504641634386)}
_TABLES = {}


def _table(filter, ratio):
    key = (filter, ratio)
    if key not in _TABLES:
        zeros, bits, rolloff, beta = _FILTERS[filter]
        n = (2 ** bits) * zeros
        win = kaiser(2 * n + 1, beta)[n:] * rolloff * np.sinc(rolloff * np.linspace(0, zeros, num=n + 1, endpoint=True))
        if ratio < 1:
            win = win * ratio
        delta = np.zeros_like(win)
        delta[:-1] = np.diff(win)
        _TABLES[key] = (np.ascontiguousarray(win), delta, 2 ** bits)
    return _TABLES[key]


def resample(x, sr_orig, sr_new, filter="kaiser_best", device="cuda:0"):
    """1-D float signal -> float64 array of int(len * sr_new / sr_orig) samples (resampy.resample semantics)"""
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
    ratio = float(sr_new) / float(sr_orig)
    n_out = int(x.shape[0] * ratio)
    if n_out < 1:
        raise ValueError(f"Input signal length={x.shape[0]} is too small to resample from {sr_orig}->{sr_new}")
    win, delta, num_table = _table(filter, ratio)
    dev = torch.device(device)
    with torch.cuda.device(dev):
        dx, dw, dd = (torch.from_numpy(a).to(dev) for a in (x, win, delta))
        y = torch.empty(n_out, dtype=torch.float64, device=dev)
        check(lib.dissc_resample(dx.data_ptr(), x.shape[0], y.data_ptr(), n_out, ratio, dw.data_ptr(), dd.data_ptr(),
                                 win.shape[0], num_table, current_stream_ptr(dev)), "dissc_resample")
        return y.cpu().numpy()


def trim(y, top_db=20, frame_length=2048, hop_length=512):
    """librosa.effects.trim semantics [3P-unverified]: keep [first, last] frames whose centred RMS is within top_db
    of the loudest frame -> (trimmed, (start, end))"""
    y = np.asarray(y, dtype=np.float64)
    half = frame_length // 2
    yp = np.concatenate([np.zeros(half), y, np.zeros(half)])
    nfr = 1 + (len(yp) - frame_length) // hop_length if len(yp) >= frame_length else 0
    if nfr <= 0:
        return y[:0], (0, 0)
    csum = np.concatenate([[0.0], np.cumsum(yp * yp)])
    starts = np.arange(nfr) * hop_length
    mse = (csum[starts + frame_length] - csum[starts]) / frame_length
    db = 10.0 * np.log10(np.maximum(1e-10, mse)) - 10.0 * np.log10(max(1e-10, mse.max()))
    keep = np.flatnonzero(db > -top_db)
    if keep.size == 0:
        return y[:0], (0, 0)
    start, end = int(keep[0] * hop_length), min(len(y), int((keep[-1] + 1) * hop_length))
    return y[start:end], (start, end)


def pad_to_multiple(data, m=1280):
    """reference data/preprocess.py:27-31"""
    if data.shape[0] % m != 0:
        data = np.pad(data, (0, m - data.shape[0] % m), mode="constant", constant_values=0)
    assert data.shape[0] % m == 0
    return data


def read_audio(path):
    """soundfile.read semantics for WAV: float64 in [-1, 1), first channel kept as is for mono; (data, sr)"""
    sr, x = wavfile.read(path)
    if x.dtype == np.int16:
        x = x.astype(np.float64) / 32768.0
    elif x.dtype == np.int32:
        x = x.astype(np.float64) / 2147483648.0
    elif x.dtype == np.uint8:
        x = (x.astype(np.float64) - 128.0) / 128.0
    else:
        x = x.astype(np.float64)
    return x, sr


def write_pcm16(path, data, sr):
    """soundfile.write(path, float data, sr) for .wav = PCM_16: libsndfile scales by 0x7FFF and rounds (its
    non-clipping float -> short conversion) [3P-unverified]; values are clipped to the int16 range here."""
    q = np.clip(np.rint(np.asarray(data, dtype=np.float64) * 32767.0), -32768, 32767).astype(np.int16)
    wavfile.write(path, sr, q)
