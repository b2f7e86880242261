"""data/preprocess.py path: resample -> trim -> pad.  CPU: the oracle restatement of resampy /
librosa.trim against analytic signals and the reference's own pad rule; the product's host pieces against the
oracle.  GPU: the HIP resampler against the oracle and the CLI end to end.  Parity with resampy / librosa
themselves is unpinned (absent offline)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
from scipy.io import wavfile

from oracle import preprocess_ref as pr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tone(sr, n, freqs=(220.0, 1330.0, 3100.0)):
    t = np.arange(n) / sr
    return sum(0.2 * np.sin(2 * np.pi * f * t + i) for i, f in enumerate(freqs))


@pytest.mark.parametrize("sr_in", [48000, 44100, 22050, 8000])
def test_oracle_resampler_reproduces_band_limited_signals(sr_in):
    n = int(0.25 * sr_in)
    freqs = (220.0, 1330.0, 3100.0)  # all below both Nyquist limits
    y = pr.resample(tone(sr_in, n, freqs), sr_in, 16000)
    assert len(y) == int(n * 16000 / sr_in)
    want = tone(16000, len(y), freqs)
    core = slice(200, len(y) - 200)  # the filter wings run off the ends of the signal
    # resampy walks its table with an INTEGER step int(scale * 512): when scale * 512 is not an integer the filter
    # is sampled slightly too densely and the pass-band gain is off by that ratio (+0.39 % for 48k -> 16k).  The
    # restatement keeps this property; allow for it here.
    scale = min(1.0, 16000 / sr_in)
    gain_err = abs(scale * 512 / int(scale * 512) - 1.0)
    assert np.abs(y[core] - want[core]).max() <= 2e-4 + 1.2 * gain_err * np.abs(want).max()
    g = np.dot(y[core], want[core]) / np.dot(want[core], want[core])
    assert abs(g - 1.0) <= 1e-4 + gain_err  # a pure gain: the waveform itself is reproduced
    assert np.abs(y[core] / g - want[core]).max() <= 1e-3


def test_oracle_trim_and_pad_rules():
    sr = 16000
    x = np.concatenate([np.zeros(8000), tone(sr, 16000), 1e-4 * np.ones(6000)])
    y, (s, e) = pr.trim(x, top_db=20)
    assert 8000 - 2048 <= s <= 8000 and 24000 <= e <= 24000 + 2048 and len(y) == e - s
    assert pr.trim(np.zeros(5000))[0].size == 5000  # all frames equal the (floored) reference level: nothing trimmed
    for n in (1, 1279, 1280, 1281, 32000, 33000):
        p = pr.pad_to_multiple(np.ones(n))
        assert len(p) % 1280 == 0 and len(p) - n < 1280 and p[:n].all() and not p[n:].any()


def test_product_host_pieces_equal_the_oracle():
    from dissc_amd import audio
    rs = np.random.RandomState(0)
    x = np.concatenate([1e-5 * rs.standard_normal(5000), 0.3 * rs.standard_normal(9000), np.zeros(7000)])
    a, sa = audio.trim(x)
    b, sb = pr.trim(x)
    assert sa == sb
    np.testing.assert_array_equal(a, b)
    for ratio, (o, n) in {"down": (48000, 16000), "up": (8000, 16000)}.items():
        win, delta, nt = audio._table("kaiser_best", n / o)
        w2, d2, nt2, r2 = pr.filter_table(o, n)
        assert nt == nt2
        np.testing.assert_allclose(win, w2, rtol=1e-13, atol=1e-18)
        np.testing.assert_allclose(delta, d2, rtol=1e-9, atol=1e-18)
    np.testing.assert_array_equal(audio.pad_to_multiple(np.arange(3.0)), pr.pad_to_multiple(np.arange(3.0)))


@pytest.mark.gpu
@pytest.mark.parametrize("sr_in", [48000, 44100, 8000])
def test_hip_resampler_matches_the_oracle(sr_in):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from dissc_amd import audio
    rs = np.random.RandomState(1)
    x = tone(sr_in, int(0.2 * sr_in)) + 0.05 * rs.standard_normal(int(0.2 * sr_in))
    want = pr.resample(x, sr_in, 16000)
    got = audio.resample(x, sr_in, 16000)
    assert got.dtype == np.float64 and got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-12
    fast = audio.resample(x, sr_in, 16000, filter="kaiser_fast")
    assert np.abs(fast - pr.resample(x, sr_in, 16000, "kaiser_fast")).max() <= 1e-12


@pytest.mark.gpu
def test_preprocess_cli(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    src, out = tmp_path / "src" / "spk", tmp_path / "out"
    src.mkdir(parents=True)
    x48 = np.concatenate([np.zeros(24000), tone(48000, 48000), np.zeros(12000)])
    wavfile.write(src / "a_001.wav", 48000, np.round(x48 * 32767).astype(np.int16))
    x16 = tone(16000, 20001)
    wavfile.write(src / "b_002.wav", 16000, np.round(x16 * 32767).astype(np.int16))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "data", "preprocess.py"), "--srcdir", str(tmp_path / "src"),
                        "--outdir", str(out), "--trim", "--pad"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert sorted(os.listdir(out)) == ["a_001.wav", "b_002.wav"]  # flat: outdir / file name, like the reference
    sr, a = wavfile.read(out / "a_001.wav")
    assert sr == 16000 and a.dtype == np.int16 and len(a) % 1280 == 0
    # oracle chain on the same int16-quantised input
    xin = np.round(x48 * 32767).astype(np.int16).astype(np.float64) / 32768.0
    want = pr.pad_to_multiple(pr.trim(pr.resample(xin, 48000, 16000))[0])
    np.testing.assert_array_equal(a, np.clip(np.rint(want * 32767.0), -32768, 32767).astype(np.int16))
    assert 16000 <= len(a) <= 16000 + 2 * 2048 + 1280  # ~1 s of tone survives the trim
    sr, b = wavfile.read(out / "b_002.wav")
    assert sr == 16000 and len(b) == 20480 and not b[20001:].any()
