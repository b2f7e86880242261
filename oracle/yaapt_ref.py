"""CPU restatement of the YAAPT pitch tracker as the reference uses it (TEST INFRASTRUCTURE).

PARITY UNPINNED.  The reference calls the un-vendored third party ``amfm_decompy.pYAAPT.yaapt`` (reference
sr/dataset.py:27-43, eval.py:26-33; inside textless' SpeechEncoder for data/encode.py:32 [3P-unverified]) with
    frame_length = 20 ms, frame_space = 5 ms, nccf_thresh1 = 0.25, tda_frame_length = 25 ms
on the float64 waveform zero-padded by 10 ms at both ends, and uses ``pitch.samp_values`` (one value per 5 ms
frame, 0.0 = unvoiced).  amfm_decompy is not installed here and there is no network, so nothing below could be
run against it: this file restates the published algorithm (S. A. Zahorian and H. Hu, "A spectral/temporal
method for robust fundamental frequency tracking", JASA 123(6), 2008) in the structure and with the default
parameter table of amfm_decompy 1.0.x AS REMEMBERED -- every function is tagged [3P-unverified].  What the
tests pin instead (tests/test_yaapt.py): known-F0 synthetic signals (sinusoid, sawtooth, harmonic complex with a
glide, voiced/unvoiced alternation) are tracked within 2 %, silence / white noise is unvoiced, and the HIP
front end (dissc_amd/f0.py) agrees with this restatement.

Stages (function names follow pYAAPT):
  bandpass       FIR order 150, 50-1500 Hz (scipy.signal.firwin, lfilter: causal, no delay compensation)
                 of the signal and of the squared ("nonlinear") signal
  nlfer          normalised low-frequency energy ratio per frame -> first voiced/unvoiced decision
  spec_track     spectral harmonics correlation (SHC) of the nonlinear signal, peak candidates, DP + smoothing
                 -> a smooth spectral F0 track and its spread
  time_track     NCCF candidates (crs_corr / cmp_rate, both signals) searched around the spectral track,
                 merits reshaped by the distance to it
  refine         merge + sort candidates, unvoiced option with merit 1 - best, spectral fall-back
  dynamic        final DP over candidates (energy-aware voiced/unvoiced transitions); unvoiced frames = 0
  f0_per_unit    textless' alignment of the 5 ms track to 20 ms units: mean of the voiced values of each
                 unit's 4 frames, 0 when none [3P-unverified]
"""
import numpy as np
from scipy import interpolate as scipy_interp
from scipy.signal import firwin, lfilter, medfilt
from scipy.signal.windows import hann, kaiser

PARAMS = {  # amfm_decompy's default table [3P-unverified], with the reference's four overrides applied
    'frame_length': 20.0, 'tda_frame_length': 25.0, 'frame_space': 5.0, 'f0_min': 60.0, 'f0_max': 400.0,
    'fft_length': 8192, 'bp_forder': 150, 'bp_low': 50.0, 'bp_high': 1500.0, 'nlfer_thresh1': 0.75,
    'nlfer_thresh2': 0.1, 'shc_numharms': 3, 'shc_window': 40.0, 'shc_maxpeaks': 4, 'shc_pwidth': 50.0,
    'shc_thresh1': 5.0, 'shc_thresh2': 1.25, 'f0_double': 150.0, 'f0_half': 150.0, 'dp5_k1': 11.0,
    'dec_factor': 1, 'nccf_thresh1': 0.25, 'nccf_thresh2': 0.9, 'nccf_maxcands': 3, 'nccf_pwidth': 5,
    'merit_boost': 0.20, 'merit_pivot': 0.99, 'merit_extra': 0.4, 'median_value': 7, 'dp_w1': 0.15,
    'dp_w2': 0.5, 'dp_w3': 0.1, 'dp_w4': 0.9, 'spec_pitch_min_std': 0.05,
}


def stride_matrix(vector, n_lin, n_col, hop):
    idx = np.arange(n_lin)[:, None] * hop + np.arange(n_col)[None, :]
    return vector[idx]


def bandpass_coeffs(fs, p=PARAMS):
    return firwin(p['bp_forder'] + 1, [p['bp_low'] / (fs / 2), p['bp_high'] / (fs / 2)], pass_zero=False)


def bandpass(x, fs, p=PARAMS):
    """causal FIR (lfilter), same length, decimation factor 1 [3P-unverified: no delay compensation]"""
    return lfilter(bandpass_coeffs(fs, p), 1.0, x)


def frame_geometry(size, fs, p=PARAMS):
    nframe = int(p['frame_length'] * fs / 1000)
    njump = int(p['frame_space'] * fs / 1000)
    samples = np.arange(int(np.fix(nframe / 2.0)), size - int(np.fix(nframe / 2.0)), njump)
    return nframe, njump, samples


def nlfer(filtered, fs, p=PARAMS):
    """-> (energy normalised by its mean [nframes], vuv bool [nframes])"""
    nfft = p['fft_length']
    nframe, njump, samples = frame_geometry(len(filtered), fs, p)
    n_f0_min = np.around((p['f0_min'] * 2 / float(fs)) * nfft)
    n_f0_max = np.around((p['f0_max'] / float(fs)) * nfft)
    window = hann(nframe + 2)[1:-1]
    frames = stride_matrix(np.asarray(filtered, dtype=np.float64), len(samples), nframe, njump) * window
    spec = np.fft.rfft(frames, nfft)
    energy = np.abs(spec[:, int(n_f0_min - 1):int(n_f0_max)]).sum(axis=1)
    mean = np.mean(energy) if len(energy) else 1.0
    energy = energy / mean if mean > 0 else energy
    return energy, energy > p['nlfer_thresh1']


def shc_geometry(fs, p=PARAMS):
    delta = fs / float(p['fft_length'])
    window_length = int(np.fix(p['shc_window'] / delta))
    half = int(np.fix(float(window_length) / 2))
    if not (window_length % 2):
        window_length += 1
    max_shc = int(np.fix((p['f0_max'] + p['shc_pwidth'] * 2) / delta))
    min_shc = int(np.ceil(p['f0_min'] / delta))
    return delta, window_length, half, min_shc, max_shc


def shc_of_magnitude(mag, fs, p=PARAMS):
    """mag: |rfft| of one frame [nfft/2+1] -> SHC [max_shc] (entries below min_shc-1 are 0)"""
    delta, wl, half, min_shc, max_shc = shc_geometry(fs, p)
    magnitude = np.zeros(half + len(mag))
    magnitude[half:] = mag
    nh = p['shc_numharms']
    prod = np.ones((max_shc - min_shc + 1, wl))
    for idx in range(nh + 1):
        prod = prod * stride_matrix(magnitude[min_shc * (idx + 1):], max_shc - min_shc + 1, wl, idx + 1)
    shc = np.zeros(max_shc)
    shc[min_shc - 1:max_shc] = prod.sum(axis=1)
    return shc


def peaks(data, delta, maxpeaks, p=PARAMS):
    """SHC peak picking -> (pitch [maxpeaks], merit [maxpeaks]); pitch 0 / merit 1 = unvoiced frame"""
    t1, t2 = p['shc_thresh1'], p['shc_thresh2']
    eps = 1e-14
    width = int(np.fix(p['shc_pwidth'] / delta))
    if not (float(width) % 2):
        width += 1
    center = int(np.ceil(width / 2.0))
    min_lag = max(int(np.fix(p['f0_min'] / delta - center)), 1)
    max_lag = min(int(np.fix(p['f0_max'] / delta + center)), len(data) - width)
    unv = (np.zeros(maxpeaks), np.ones(maxpeaks))
    max_data = np.max(data[min_lag:max_lag + 1])
    if max_data > eps:
        data = data / max_data
    avg = np.mean(data[min_lag:max_lag + 1])
    if avg > 1.0 / t1:
        return unv
    seg = slice(min_lag + center + 1, max_lag - center + 1)
    cur = data[seg]
    cand = (cur > data[min_lag + center:max_lag - center]) & (cur > data[min_lag + center + 2:max_lag - center + 2]) \
        & (cur > t2 * avg)
    pitch, merit = [], []
    for n in (np.nonzero(cand)[0] + min_lag + center + 1).tolist():
        if np.argmax(data[n - center:n + center + 1]) == center:
            pitch.append(float(n) * delta)
            merit.append(data[n])
    if not pitch or max(merit) / avg < t1:
        return unv
    order = np.argsort(-np.asarray(merit), kind="stable")
    n = min(len(pitch), maxpeaks)
    pitch = np.append(np.asarray(pitch)[order][:n], np.zeros(maxpeaks - n))
    merit = np.append(np.asarray(merit)[order][:n], np.zeros(maxpeaks - n))
    # step 4: insert candidates against pitch doubling / halving, with the merit of the SECOND peak
    if pitch[0] > p['f0_double']:
        n = min(n + 1, maxpeaks)
        pitch[n - 1], merit[n - 1] = pitch[0] / 2.0, merit[1]
    if pitch[0] < p['f0_half']:
        n = min(n + 1, maxpeaks)
        pitch[n - 1], merit[n - 1] = pitch[0] * 2.0, merit[1]
    # step 5: pad with the best candidate
    if n < maxpeaks:
        pitch[n:], merit[n:] = pitch[0], merit[0]
    return pitch, merit


def dynamic5(pitch_array, merit_array, k1, f0_min):
    """DP over [cands, frames]: local cost 1 - merit, transition k1 * |df| / f0_min -> best pitch per frame"""
    ncands, nframes = pitch_array.shape
    local = 1.0 - merit_array
    prev = np.zeros((ncands, nframes), dtype=int)
    cum = local[:, 0].copy()
    for i in range(1, nframes):
        trans = k1 * np.abs(pitch_array[:, i][:, None] - pitch_array[:, i - 1][None, :]) / f0_min  # [j, k]
        tot = cum[None, :] + trans
        prev[:, i] = np.argmin(tot, axis=1)
        cum = tot[np.arange(ncands), prev[:, i]] + local[:, i]
    path = np.zeros(nframes, dtype=int)
    path[-1] = int(np.argmin(cum))
    for i in range(nframes - 1, 0, -1):
        path[i - 1] = prev[path[i], i]
    return pitch_array[path, np.arange(nframes)]


def spec_track(nl_filtered, fs, vuv, p=PARAMS):
    """-> (spectral F0 track [nframes], its spread, vuv possibly narrowed)"""
    nfft = p['fft_length']
    nframe, njump, samples = frame_geometry(len(nl_filtered), fs, p)
    nframes = len(samples)
    nframe2 = nframe * 2
    maxpeaks = p['shc_maxpeaks']
    delta = fs / float(nfft)
    cand_pitch = np.zeros((maxpeaks, nframes))
    cand_merit = np.ones((maxpeaks, nframes))
    need = nframe2 + (nframes - 1) * njump
    data = np.append(nl_filtered, np.zeros(max(0, need - len(nl_filtered))))
    window = kaiser(nframe2, 0.5)
    for frame in np.nonzero(vuv)[0].tolist():
        s = data[frame * njump:frame * njump + nframe2] * window
        s = s - np.mean(s)
        shc = shc_of_magnitude(np.abs(np.fft.rfft(s, nfft)), fs, p)
        cand_pitch[:, frame], cand_merit[:, frame] = peaks(shc, delta, maxpeaks, p)
    return spec_track_from_candidates(cand_pitch, cand_merit, p)


def spec_track_from_candidates(cand_pitch, cand_merit, p=PARAMS):
    nframes = cand_pitch.shape[1]
    spec_pitch = cand_pitch[0].copy()
    vmask = cand_pitch[0] > 0
    vp, vm = cand_pitch[:, vmask].copy(), cand_merit[:, vmask].copy()
    nv = vp.shape[1]
    if nv == 0:
        return np.full(nframes, 150.0), 150.0 * p['spec_pitch_min_std'], vmask
    avg_v, std_v = np.mean(vp[0]), np.std(vp[0])
    delta1 = np.abs(vp - 0.8 * avg_v) * (3 - vm)
    index = delta1.argmin(0)
    cols = np.arange(nv)
    peak_min = medfilt(vp[index, cols], max(1, p['median_value'] - 2))
    merit_min = vm[index, cols]
    vp[index, cols] = peak_min
    vm[index, cols] = merit_min
    weight_trans = p['dp5_k1'] * std_v / avg_v
    if nv > 2:
        voiced_pitch = dynamic5(vp, vm, weight_trans, p['f0_min'])
        voiced_pitch = medfilt(voiced_pitch, max(1, p['median_value'] - 2))
    else:
        voiced_pitch = np.full(nv, 150.0)
    pitch_avg = np.mean(voiced_pitch)
    pitch_std = max(np.std(voiced_pitch), pitch_avg * p['spec_pitch_min_std'])
    spec_pitch[vmask] = voiced_pitch
    if spec_pitch[0] < pitch_avg / 2:
        spec_pitch[0] = pitch_avg
    if spec_pitch[-1] < pitch_avg / 2:
        spec_pitch[-1] = pitch_avg
    nz = np.nonzero(spec_pitch)[0]
    if len(nz) > 1:
        spec_pitch = scipy_interp.pchip(nz, spec_pitch[nz])(np.arange(nframes))
    else:
        spec_pitch = np.full(nframes, pitch_avg)
    spec_pitch = lfilter(np.ones(3) / 3, 1.0, spec_pitch)
    if nframes > 3:
        spec_pitch[0], spec_pitch[1] = spec_pitch[2], spec_pitch[3]
    return spec_pitch, pitch_std, vmask


def crs_corr(frame, lag_min, lag_max):
    """NCCF of one frame (mean removed): phi[lag] = <x[0:N], x[lag:lag+N]> / sqrt(<x0,x0><xl,xl>) for
    lag_min <= lag < lag_max, N = len(frame) - lag_max (the correlation window shrinks with the largest lag)"""
    data = np.asarray(frame, dtype=np.float64)
    data = data - np.mean(data)
    n = len(data) - lag_max
    assert n > 0
    phi = np.zeros(len(data))
    x_j = data[:n]
    e0 = np.dot(x_j, x_j)
    m = stride_matrix(data[lag_min:lag_max + n], lag_max - lag_min, n, 1)
    den = np.sum(m * m, axis=1) * e0
    num = m @ x_j
    phi[lag_min:lag_max] = np.where(den > 0, num / np.sqrt(np.where(den > 0, den, 1.0)), 0.0)
    return phi


def cmp_rate(phi, fs, maxcands, lag_min, lag_max, p=PARAMS):
    """NCCF peaks -> (pitch [maxcands], merit [maxcands]); pitch = fs / lag.  (phi is indexed by the lag itself
    here; whether amfm_decompy divides by lag or lag + 1 could not be checked -- the unbiased form is used, it
    is the one that tracks known-F0 signals to < 1 %.)"""
    width = p['nccf_pwidth']
    center = int(np.fix(width / 2.0))
    t1, t2 = p['nccf_thresh1'], p['nccf_thresh2']
    a, b = lag_min + center, lag_max - center + 1
    cur = phi[a:b]
    cand = (cur > phi[a - 1:b - 1]) & (cur > phi[a + 1:b + 1]) & (cur > t1)
    pk = (np.nonzero(cand)[0] + a).tolist()
    pitch, merit = [], []
    if pk and np.amax(phi) > t2:
        best = pk[int(np.argmax(phi[pk]))]
        pitch, merit = [fs / float(best)], [phi[best]]
    else:
        for n in pk:
            if np.argmax(phi[n - center:n + center + 1]) == center:
                pitch.append(fs / float(n))
                merit.append(phi[n])
    if not pitch:
        return np.zeros(maxcands), np.full(maxcands, 0.001)
    order = np.argsort(-np.asarray(merit), kind="stable")[:maxcands]
    pp, mm = np.asarray(pitch)[order], np.asarray(merit)[order]
    if len(pp) < maxcands:
        mm = np.append(mm, np.full(maxcands - len(pp), mm[0]))
        pp = np.append(pp, np.full(maxcands - len(pp), pp[0]))
    return pp, mm


def tda_geometry(size, fs, nframes_spec, p=PARAMS):
    n = int(p['tda_frame_length'] * fs / 1000)
    njump = int(p['frame_space'] * fs / 1000)
    nfr = int((size - (n - njump)) / njump)
    return n, njump, min(nfr, nframes_spec)


def lag_ranges(spec_pitch, pitch_std, fs, p=PARAMS):
    lo = np.maximum(spec_pitch - 2.0 * pitch_std, p['f0_min'])
    hi = np.minimum(spec_pitch + 2.0 * pitch_std, p['f0_max'])
    half = int(np.fix(p['nccf_pwidth'] / 2.0))
    return np.fix(fs / hi).astype(int) - half, np.fix(fs / lo).astype(int) + half


def time_track(filtered, fs, spec_pitch, pitch_std, p=PARAMS):
    """NCCF candidates per frame; lags restricted to spec_pitch -+ 2 std; merits reshaped by the distance to
    the spectral track: merit = (1 + boost) * merit * max(0, 1 - |f - f_spec| / (5 std))"""
    n, njump, nfr = tda_geometry(len(filtered), fs, len(spec_pitch), p)
    maxc = p['nccf_maxcands']
    spec = spec_pitch[:nfr]
    lmin, lmax = lag_ranges(spec, pitch_std, fs, p)
    data = np.asarray(filtered, dtype=np.float64)
    pitch = np.zeros((maxc, nfr))
    merit = np.zeros((maxc, nfr))
    for f in range(nfr):
        phi = crs_corr(data[f * njump:f * njump + n], int(lmin[f]), int(lmax[f]))
        pitch[:, f], merit[:, f] = cmp_rate(phi, fs, maxc, int(lmin[f]), int(lmax[f]), p)
    return reshape_merit(pitch, merit, spec, pitch_std, p)


def reshape_merit(pitch, merit, spec, pitch_std, p=PARAMS):
    thresh = 5.0 * pitch_std
    diff = np.abs(pitch - spec[None, :])
    match = np.where(diff < thresh, 1.0 - diff / thresh, 0.0)
    return pitch, (1.0 + p['merit_boost']) * merit * match


def refine(tp1, tm1, tp2, tm2, spec_pitch, energy, vuv, p=PARAMS):
    """merge the two candidate sets, sort by merit, give every frame an unvoiced option, fall back to the
    spectral estimate where the NCCF found nothing, add the smoothed best track and the spectral track"""
    nfr = min(tp1.shape[1], tp2.shape[1])
    spec, en, vu = spec_pitch[:nfr], energy[:nfr], vuv[:nfr]
    pitch = np.vstack([tp1[:, :nfr], tp2[:, :nfr]])
    merit = np.vstack([tm1[:, :nfr], tm2[:, :nfr]])
    order = np.argsort(-merit, axis=0, kind="stable")
    cols = np.arange(nfr)[None, :]
    pitch, merit = pitch[order, cols], merit[order, cols]
    K = pitch.shape[0]
    best_pitch = medfilt(pitch[0], p['median_value']) * vu
    for i in range(nfr):
        if en[i] <= p['nlfer_thresh2']:      # definitely unvoiced
            pitch[:, i], merit[:, i] = 0.0, p['merit_pivot']
        elif pitch[0, i] > 0:                # voiced candidate present: make the last one the unvoiced option
            pitch[K - 1, i], merit[K - 1, i] = 0.0, 1.0 - merit[0, i]
            for j in range(1, K - 1):
                if pitch[j, i] == 0:
                    merit[j, i] = 0.0
        else:                                # nothing from the NCCF: the spectral estimate, merit from the energy
            pitch[0, i], merit[0, i] = spec[i], min(1.0, en[i] / 2.0)
            pitch[1:, i], merit[1:, i] = 0.0, 1.0 - merit[0, i]
    # two extra rows: the smoothed best track and the spectral track (both only on NLFER-voiced frames)
    extra_p = np.vstack([best_pitch, spec * vu])
    extra_m = np.vstack([np.where(best_pitch > 0, p['merit_extra'], 0.0), np.where(vu, p['merit_extra'], 0.0)])
    return np.vstack([pitch[:K - 1], extra_p, pitch[K - 1:]]), np.vstack([merit[:K - 1], extra_m, merit[K - 1:]])


def dynamic(ref_pitch, ref_merit, energy, p=PARAMS):
    """final DP (Zahorian & Hu eqs. 17-21): local cost w4 * (1 - merit); transitions: voiced-voiced
    w1 * |df| / mean F0, voiced<->unvoiced w2 * (1 - min(1, |dE|)), unvoiced-unvoiced w3"""
    K, nfr = ref_pitch.shape
    en = energy[:nfr]
    best = ref_pitch[0]
    mean_pitch = np.mean(best[best > 0]) if np.any(best > 0) else 150.0
    local = p['dp_w4'] * (1.0 - ref_merit)
    cum = local[:, 0].copy()
    prev = np.zeros((K, nfr), dtype=int)
    for i in range(1, nfr):
        pj, pk = ref_pitch[:, i][:, None], ref_pitch[:, i - 1][None, :]
        both = (pj > 0) & (pk > 0)
        neither = (pj == 0) & (pk == 0)
        benefit = min(1.0, abs(en[i - 1] - en[i]))
        trans = np.where(both, p['dp_w1'] * np.abs(pj - pk) / mean_pitch,
                         np.where(neither, p['dp_w3'], p['dp_w2'] * (1.0 - benefit)))
        tot = cum[None, :] + trans
        prev[:, i] = np.argmin(tot, axis=1)
        cum = tot[np.arange(K), prev[:, i]] + local[:, i]
    path = np.zeros(nfr, dtype=int)
    path[-1] = int(np.argmin(cum))
    for i in range(nfr - 1, 0, -1):
        path[i - 1] = prev[path[i], i]
    return ref_pitch[path, np.arange(nfr)]


def yaapt(signal, fs=16000, p=PARAMS):
    """float waveform -> per-frame F0 (Hz, 0 = unvoiced), one value per frame_space (pYAAPT samp_values)"""
    x = np.asarray(signal, dtype=np.float64)
    filt = bandpass(x, fs, p)
    nl_filt = bandpass(x * x, fs, p)
    energy, vuv = nlfer(filt, fs, p)
    if len(energy) == 0:
        return np.zeros(0)
    spec_pitch, pitch_std, _ = spec_track(nl_filt, fs, vuv, p)
    tp1, tm1 = time_track(filt, fs, spec_pitch, pitch_std, p)
    tp2, tm2 = time_track(nl_filt, fs, spec_pitch, pitch_std, p)
    rp, rm = refine(tp1, tm1, tp2, tm2, spec_pitch, energy, vuv, p)
    f0 = dynamic(rp, rm, energy, p)
    out = np.zeros(len(energy))  # frames beyond the last full time-domain frame stay unvoiced
    out[:len(f0)] = f0
    return out


def get_yaapt_f0(audio, rate=16000):
    """reference sr/dataset.py:27-43 (interp=False): pad 10 ms of zeros at both ends, track"""
    to_pad = int(20.0 / 1000 * rate) // 2
    y = np.pad(np.asarray(audio, dtype=np.float64).reshape(-1), (to_pad, to_pad), "constant")
    return yaapt(y, rate)


def f0_per_unit(f0_frames, n_units, ratio=4):
    """textless align_f0_to_durations for deduplicate=False [3P-unverified]: the track is truncated to
    ratio * n_units frames or, when shorter, extended with its last value; unit i covers frames
    [ratio*i, ratio*(i+1)); mean of its voiced (non-zero) values, 0.0 if none"""
    f0_frames = np.asarray(f0_frames, dtype=np.float64)[:ratio * n_units]
    if 0 < len(f0_frames) < ratio * n_units:
        f0_frames = np.concatenate([f0_frames, np.full(ratio * n_units - len(f0_frames), f0_frames[-1])])
    out = np.zeros(n_units)
    for i in range(n_units):
        seg = np.asarray(f0_frames[ratio * i:ratio * (i + 1)])
        seg = seg[seg != 0]
        out[i] = seg.mean() if len(seg) else 0.0
    return out
