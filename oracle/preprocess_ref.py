"""CPU restatement of the audio preparation step before the path (TEST INFRASTRUCTURE): reference
data/preprocess.py:19-36 = soundfile read -> resampy.resample(data, sr, 16000) -> optional
librosa.effects.trim(top_db=20) -> optional zero padding to a multiple of 1280 samples -> soundfile write.

PARITY UNPINNED for the two third-party pieces (resampy, librosa: not installed, no network); both are
restated from their published algorithms AS REMEMBERED [3P-unverified].  Pinned here: the pad rule (pure
numpy in the reference), resampling against analytic band-limited signals, and the HIP kernel against this file.

  sinc_window / resample   <- resampy.filters.sinc_window, resampy.core.resample + interpn.resample_f
                              ('kaiser_best': 64 zero crossings, 2^9 table entries each, roll-off
                              0.9475937167399596, Kaiser beta 14.769656459379492; 'kaiser_fast': 16 / 2^9 /
                              0.85 / 8.555504641634386)
  trim                     <- librosa.effects.trim (frame 2048, hop 512, centred RMS, dB relative to the peak)
  pad_to_multiple          <- reference data/preprocess.py:27-31
"""
import numpy as np
from scipy.signal.windows import kaiser

FILTERS = {"kaiser_best": dict(num_zeros=64, precision=9, rolloff=0.9475937167399596, beta=14.769656459379492),
           "kaiser_fast": dict(num_zeros=16, precision=9, rolloff=0.85, beta=8.555504641634386)}


def sinc_window(num_zeros, precision, rolloff, beta):
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


def filter_table(sr_orig, sr_new, filter="kaiser_best"):
    """-> (interp_win, interp_delta, num_table, ratio): the table resample_f walks (scaled when downsampling)"""
    ratio = float(sr_new) / sr_orig
    win, num_table = sinc_window(**FILTERS[filter])
    win = win.copy()
    if ratio < 1:
        win *= ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    return win, delta, num_table, ratio


def resample(x, sr_orig, sr_new, filter="kaiser_best"):
    """1-D float64 signal -> int(len * ratio) samples"""
    x = np.asarray(x, dtype=np.float64)
    win, delta, num_table, ratio = filter_table(sr_orig, sr_new, filter)
    n_out = int(x.shape[0] * ratio)
    y = np.zeros(n_out)
    scale = min(1.0, ratio)
    index_step = int(scale * num_table)
    nwin = win.shape[0]
    t_out = np.arange(n_out) * (1.0 / ratio)
    for t in range(n_out):
        tr = t_out[t]
        n = int(tr)
        frac = scale * (tr - n)
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        i_max = min(n + 1, (nwin - offset) // index_step)
        idx = offset + np.arange(i_max) * index_step
        y[t] += np.dot(win[idx] + eta * delta[idx], x[n - np.arange(i_max)])
        frac = scale - frac
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        k_max = min(x.shape[0] - n - 1, (nwin - offset) // index_step)
        idx = offset + np.arange(k_max) * index_step
        y[t] += np.dot(win[idx] + eta * delta[idx], x[n + 1 + np.arange(k_max)])
    return y


def trim(y, top_db=20, frame_length=2048, hop_length=512):
    """-> (trimmed signal, (start, end)): frames whose RMS is within top_db of the loudest frame"""
    y = np.asarray(y, dtype=np.float64)
    pad = frame_length // 2
    yp = np.pad(y, (pad, pad), mode="constant")
    n_frames = 1 + (len(yp) - frame_length) // hop_length if len(yp) >= frame_length else 0
    if n_frames <= 0:
        return y[:0], (0, 0)
    idx = np.arange(n_frames)[:, None] * hop_length + np.arange(frame_length)[None, :]
    mse = np.mean(yp[idx] ** 2, axis=1)
    ref = mse.max()
    amin = 1e-10
    db = 10.0 * np.log10(np.maximum(amin, mse)) - 10.0 * np.log10(np.maximum(amin, ref))
    nz = np.nonzero(db > -top_db)[0]
    if len(nz) == 0:
        return y[:0], (0, 0)
    start = int(nz[0] * hop_length)
    end = min(len(y), int((nz[-1] + 1) * hop_length))
    return y[start:end], (start, end)


def pad_to_multiple(data, m=1280):
    """reference data/preprocess.py:27-31"""
    if data.shape[0] % m != 0:
        data = np.pad(data, (0, m - data.shape[0] % m), mode="constant", constant_values=0)
    assert data.shape[0] % m == 0
    return data
