"""Predictor training: the CPU restatement of one optimisation step (oracle/train_ref.py) against
outputs of the REFERENCE modules themselves in train() mode (tests/golden/train.npz, made by
tests/golden/make_golden.py `train`: reference LenPredictor / PitchPredictor / PitchPredictorBase + LenSumLoss /
PitchLoss + torch.optim.Adam, two steps, the random masks injected)."""
import os

import numpy as np
import pytest
import torch

import synthdata as synth
from oracle import train_ref as tr

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def compact(a):
    a = np.asarray(a)
    if a.size <= 4096:
        return a.copy()
    f = a.reshape(-1).astype(np.float64)
    return np.concatenate([[f.sum(), np.abs(f).sum(), (f * f).sum()], f[::max(1, f.size // 509)]])


def close(got, want, rtol, atol, what=""):
    got, want = compact(got), np.asarray(want)
    assert got.shape == want.shape, what
    scale = np.abs(want).max() if want.size else 0.0
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol + rtol * scale, err_msg=what)


BN_FED_BIASES = {"len": {"cnn1.bias"} | {f"cnn1{i}.bias" for i in range(1, 7)},
                 "new": {"cnn2.bias"},
                 "base": {"cnn1.bias", "cnn_class1.bias", "cnn_reg1.bias"} | {f"cnn1{i}.bias" for i in range(1, 8)}}


def initial_state(kind, n_spk=108):
    if kind == "len":
        return synth.synth_len_state_dict(100, n_spk)
    return synth.synth_pitch_state_dict(kind, 100, n_spk)


@pytest.mark.parametrize("kind", ["len", "new", "base"])
def test_training_step_oracle_matches_the_reference(kind):
    g = np.load(os.path.join(GOLDEN, "train.npz"))
    sd = {k: v.clone() for k, v in initial_state(kind).items()}
    stats = (torch.from_numpy(g["id2pitch_mean"]), torch.from_numpy(g["id2pitch_std"]))
    state = {}
    for step in range(2):
        pre = f"{kind}/s{step}/"
        seq, tgt, spk, keep = (torch.from_numpy(g[pre + n]) for n in ("seq", "tgt", "spk", "keep"))
        pe_mult = torch.from_numpy(g[pre + "pe_mult"]) if kind == "new" else None
        loss, grads = tr.train_step(kind, sd, seq, spk, tgt, keep, float(g[f"{kind}/lr"]), state,
                                    norm=(torch.tensor(3.3), torch.tensor(2.1)), stats=stats, pe_mult=pe_mult)
        assert abs(float(loss) - float(g[pre + "loss"])) <= 2e-5 * abs(float(g[pre + "loss"]))
        lr = float(g[f"{kind}/lr"])
        for k in tr.trainable_keys(sd):
            if k in BN_FED_BIASES[kind]:  # pure rounding noise on both sides (see below): only its size is checked
                wscale = np.abs(compact(grads[k[:-4] + "weight"].numpy())[3:]).max()
                assert np.abs(grads[k].numpy()).max() <= 1e-3 * wscale
                assert np.abs(g[pre + "grad/" + k]).max() <= 1e-3 * wscale
                continue
            if step > 0 and kind != "len":
                # PitchLoss is |.| + BCE: its gradient is discontinuous where a prediction crosses its target, so the
                # 1e-5-level parameter differences left by step 0 flip a few per-frame signs in step 1 (an O(1) change
                # for those frames).  Second-step gradients of the pitch models are compared in the l2 sense.
                a, b = compact(grads[k].numpy()).astype(np.float64), np.asarray(g[pre + "grad/" + k], dtype=np.float64)
                assert np.linalg.norm(a - b) <= 0.05 * np.linalg.norm(b) + 1e-6, f"grad {k} step {step}"
                continue
            close(grads[k].numpy(), g[pre + "grad/" + k], 2e-4, 5e-6, f"grad {k} step {step}")
        for k, v in sd.items():
            # A conv bias in front of a BatchNorm has an exactly-zero gradient in exact arithmetic (the batch mean
            # is subtracted again): what reaches Adam is rounding noise of ~1e-7, which Adam normalises to a full
            # +-lr step of random sign -- in the reference as well.  Those parameters are compared to within the
            # random walk they perform; everything else tightly.
            noise = 2.2 * lr * (step + 1) if k in BN_FED_BIASES[kind] else 0.0
            if k.endswith("running_mean"):  # the batch mean moves with that bias: momentum 0.1 of its random walk
                noise = 0.1 * 2.2 * lr * step
            if step > 0 and kind != "len" and k in tr.trainable_keys(sd):
                noise = max(noise, 2.2 * lr)  # a flipped gradient sign moves Adam's step by up to 2 lr
            close(v.numpy(), g[pre + "after/" + k], 2e-5 if step == 0 else 2e-4, 1e-7 + noise, f"after {k} step {step}")
    # padding rows never move: token_emb's pad row has no gradient
    assert torch.equal(sd["token_emb.weight"][100], initial_state(kind)["token_emb.weight"][100])


def test_long_batch_step_oracle_matches_the_reference():
    """the second length-model case of the golden file: 3 x 150 padded units (longer than a 128-column tile, no
    multiple of 64), every LeakyReLU input >= 2e-5 away from 0 (the generator of the fixture searches the batch seed)"""
    g = np.load(os.path.join(GOLDEN, "train.npz"))
    assert float(g["len_long/min_abs_preact"]) >= 2e-5 and float(g["len/s0/min_abs_preact"]) >= 2e-5
    sd = {k: v.clone() for k, v in initial_state("len").items()}
    seq, tgt, spk, keep = (torch.from_numpy(g["len_long/" + n]) for n in ("seq", "tgt", "spk", "keep"))
    assert seq.shape == (3, 150)
    loss, grads = tr.train_step("len", sd, seq, spk, tgt, keep, 3e-4, {}, norm=(torch.tensor(3.3), torch.tensor(2.1)))
    assert abs(float(loss) - float(g["len_long/loss"])) <= 2e-5 * abs(float(g["len_long/loss"]))
    for k in tr.trainable_keys(sd):
        if k in BN_FED_BIASES["len"]:
            continue
        close(grads[k].numpy(), g["len_long/grad/" + k], 2e-4, 5e-6, f"grad {k}")
