"""GPU parity of the HuBERT unit encoder vs the HF/sklearn goldens and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from dissc_amd.hubert import HubertEncoder
    from oracle import hubert_ref as hr
    import synthdata as synth
    sd = synth.synth_hubert_state_dict(6)
    centers = synth.synth_kmeans_centers()
    enc = HubertEncoder(sd, centers, n_layers=6).to("cuda:0")
    return dict(enc=enc, sd=sd, centers=centers, hr=hr, synth=synth,
                g=np.load(os.path.join(golden_dir, "hubert.npz")))


def _check_units(units, dense_ref, centers, want):
    d = ((torch.from_numpy(dense_ref)[:, None, :] - centers[None]) ** 2).sum(-1)
    top2 = torch.topk(d, 2, largest=False).values
    safe = ((top2[:, 1] - top2[:, 0]) > 0.02).numpy()
    np.testing.assert_array_equal(units[safe], want[safe])
    return int((~safe).sum()), int((units != want).sum())


@pytest.mark.parametrize("n", [400, 719, 4000, 16000, 32000])
def test_hubert_matches_hf_golden(env, n):
    g = env["g"]
    wav = torch.from_numpy(env["synth"].synth_waveform(n, seed=n))[None]
    out = env["enc"](wav)
    dense = out["dense"][0].cpu().numpy()
    want = g[f"n{n}/dense"]
    assert dense.shape == want.shape
    err = np.abs(dense - want).max()
    assert err <= 5e-4 * max(1.0, np.abs(want).max()), err
    near, diff = _check_units(out["units"][0].cpu().numpy(), want, env["centers"], g[f"n{n}/units"])
    assert diff <= near  # indices may differ only at near-ties


def test_hubert_ragged_batch_is_per_utterance_exact(env):
    ns = [32000, 16000, 4000, 719]
    N = max(ns)
    wav = torch.zeros(len(ns), N)
    for i, n in enumerate(ns):
        wav[i, :n] = torch.from_numpy(env["synth"].synth_waveform(n, seed=n))
        wav[i, n:] = float("nan")  # padding must never be read
    out = env["enc"](wav, n_samples=torch.tensor(ns))
    for i, n in enumerate(ns):
        T = int(out["frames"][i])
        one = env["enc"](wav[i:i + 1, :n])
        assert int(one["frames"][0]) == T
        np.testing.assert_array_equal(out["units"][i, :T].cpu().numpy(), one["units"][0].cpu().numpy())
        a, b = out["dense"][i, :T].cpu().numpy(), one["dense"][0].cpu().numpy()
        assert np.isfinite(a).all()
        assert np.abs(a - b).max() <= 1e-5  # same kernels; only tile partitioning may differ
        want = env["g"][f"n{n}/dense"]
        assert np.abs(a - want).max() <= 5e-4 * max(1.0, np.abs(want).max())


def test_hubert_10s_against_oracle(env):
    n = 160000
    wav = torch.from_numpy(env["synth"].synth_waveform(n, seed=1))[None]
    out = env["enc"](wav)
    assert out["units"].shape == (1, 499)
    units_ref, dense_ref = env["hr"].encode(env["sd"], env["centers"], wav)
    err = np.abs(out["dense"][0].cpu().numpy() - dense_ref.numpy()).max()
    assert err <= 5e-4 * max(1.0, float(dense_ref.abs().max())), err
    near, diff = _check_units(out["units"][0].cpu().numpy(), dense_ref.numpy(), env["centers"], units_ref.numpy())
    assert diff <= near
