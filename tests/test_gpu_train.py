"""Predictor training on the GPU: one optimisation step of dissc_amd.train.Trainer (csrc/train.hip
through the C ABI) against the REFERENCE's own training step (tests/golden/train.npz: reference modules in train()
mode + LenSumLoss / PitchLoss + torch.optim.Adam, masks injected) and against the CPU oracle; then short training
runs and the two training CLIs."""
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

from test_train_oracle import BN_FED_BIASES, compact, initial_state

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def gold():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return np.load(os.path.join(GOLDEN, "train.npz"))


def _close(got, want, rtol, atol, what):
    got, want = compact(got), np.asarray(want)
    assert got.shape == want.shape, what
    scale = np.abs(want).max() if want.size else 0.0
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol + rtol * scale, err_msg=what)


@pytest.mark.parametrize("kind", ["len", "new", "base"])
def test_training_step_matches_the_reference(gold, kind):
    from dissc_amd.train import Trainer
    g = gold
    lr = float(g[f"{kind}/lr"])
    stats = (torch.from_numpy(g["id2pitch_mean"]), torch.from_numpy(g["id2pitch_std"]))
    tr = Trainer(kind, initial_state(kind), lr, norm=(3.3, 2.1), stats=stats).to("cuda:0")
    for step in range(2):
        pre = f"{kind}/s{step}/"
        seq, tgt, spk, keep = (torch.from_numpy(g[pre + n]) for n in ("seq", "tgt", "spk", "keep"))
        pe_mult = torch.from_numpy(g[pre + "pe_mult"]) if kind == "new" else None
        loss = float(tr.step(seq, spk, tgt, keep=keep, pe_mult=pe_mult))
        assert abs(loss - float(g[pre + "loss"])) <= 5e-5 * abs(float(g[pre + "loss"])), (loss, float(g[pre + "loss"]))
        grads = tr.grads()
        for k, gv in grads.items():
            want = g[pre + "grad/" + k]
            if k in BN_FED_BIASES[kind]:  # exactly zero in exact arithmetic: rounding noise on both sides
                wscale = np.abs(compact(grads[k[:-4] + "weight"].numpy())[3:]).max()
                assert np.abs(gv.numpy()).max() <= 1e-3 * wscale, k
                continue
            if step > 0:
                # second step of the SAME run: the weights already differ by Adam's +-lr sign flips (and, for the pitch
                # models, the loss gradient is discontinuous, see tests/test_train_oracle.py) -- l2 sanity only; the
                # exact second-step check is test_second_step_from_the_oracles_state below
                a, b = compact(gv.numpy()).astype(np.float64), np.asarray(want, dtype=np.float64)
                assert np.linalg.norm(a - b) <= (0.15 if kind == "len" else 0.05) * np.linalg.norm(b) + 1e-6, k
                continue
            # LeakyReLU's derivative jumps at 0: an activation within rounding of 0 would take the other branch and
            # move every gradient below it by O(1 %) (round 2's length batch had one, in channel 104 of the last trunk
            # layer).  The golden batches are now generated with every pre-activation >= 2e-5 away from 0
            # (make_golden.py _search_len_batch), so all three models are held to 1e-4 in the l2 sense.
            a, b = compact(gv.numpy()).astype(np.float64), np.asarray(want, dtype=np.float64)
            lim = 1e-4
            assert np.linalg.norm(a - b) <= lim * np.linalg.norm(b) + 1e-6, \
                (f"grad {k} step {step}", np.linalg.norm(a - b) / np.linalg.norm(b))
        sd = tr.state_dict()
        assert list(sd) == list(initial_state(kind))
        for k, v in sd.items():
            noise = 2.2 * lr * (step + 1) if k in BN_FED_BIASES[kind] else 0.0
            if k.endswith("running_mean"):
                noise = 0.1 * 2.2 * lr * step
            if step > 0 and kind != "len" and k in grads:
                noise = max(noise, 2.2 * lr)
            if kind == "len" and k in grads:
                # Adam's normalised step: an element whose gradient is within the 1.4 % above of zero moves by
                # +lr instead of -lr
                noise = max(noise, 2.2 * lr * (step + 1))
            if k.endswith("running_var") or k.endswith("running_mean"):
                noise = max(noise, 1e-5 if step == 0 or kind != "len" else 2 * lr)
            _close(v.float().numpy(), g[pre + "after/" + k], 1e-4, 1e-6 + noise, f"after {k} step {step}")
        nbt = [k for k in sd if k.endswith("num_batches_tracked")][0]
        assert int(sd[nbt]) == int(initial_state(kind)[nbt]) + step + 1


def test_long_batch_gradients_match_the_reference(gold):
    """3 x 150 padded units: longer than the 128-column tiles of the conv / weight-gradient kernels and no multiple of
    64, so the time split of train_wgrad_mfma_kernel has its seam inside every utterance; every gradient of the
    length model against the reference's at 1e-4 (l2) and elementwise."""
    from dissc_amd.train import Trainer
    g = gold
    tr = Trainer("len", initial_state("len"), 3e-4, norm=(3.3, 2.1)).to("cuda:0")
    seq, tgt, spk, keep = (torch.from_numpy(g["len_long/" + n]) for n in ("seq", "tgt", "spk", "keep"))
    loss = float(tr.step(seq, spk, tgt, keep=keep))
    assert abs(loss - float(g["len_long/loss"])) <= 5e-5 * abs(float(g["len_long/loss"]))
    for k, gv in tr.grads().items():
        if k in BN_FED_BIASES["len"]:
            continue
        a, b = compact(gv.numpy()).astype(np.float64), np.asarray(g["len_long/grad/" + k], dtype=np.float64)
        assert np.linalg.norm(a - b) <= 1e-4 * np.linalg.norm(b) + 1e-6, (k, np.linalg.norm(a - b) / np.linalg.norm(b))
        _close(gv.numpy(), g["len_long/grad/" + k], 2e-4, 5e-6, f"grad {k}")


def _flip_tolerant(got, want, what, l2=0.03):
    """LeakyReLU's derivative jumps at 0, so an activation within fp32 rounding of 0 takes the other branch in one of
    the two implementations and perturbs the gradients that pass through it (a few rows, O(1 %) of the norm).  A real
    defect moves everything: require the typical (median) element to agree tightly and the whole tensor in the l2 sense."""
    a, b = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert a.shape == b.shape, what
    assert np.median(np.abs(a - b)) <= 1e-3 * np.abs(b).max() + 1e-6, (what, float(np.median(np.abs(a - b))))
    assert np.linalg.norm(a - b) <= l2 * np.linalg.norm(b) + 1e-6, (what, np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.mark.parametrize("kind", ["len", "new", "base"])
def test_second_step_from_the_oracles_state(gold, kind):
    """the engine loaded with the weights the oracle holds after its first optimisation step, given the second golden
    batch: loss and every gradient against the oracle's (which tests/test_train_oracle.py pins to the reference)"""
    from oracle import train_ref as tr_ref
    from dissc_amd.train import Trainer
    g = gold
    lr = float(g[f"{kind}/lr"])
    stats = (torch.from_numpy(g["id2pitch_mean"]), torch.from_numpy(g["id2pitch_std"]))
    norm = (torch.tensor(3.3), torch.tensor(2.1))
    sd, state = {k: v.clone() for k, v in initial_state(kind).items()}, {}
    for step in range(2):
        pre = f"{kind}/s{step}/"
        seq, tgt, spk, keep = (torch.from_numpy(g[pre + n]) for n in ("seq", "tgt", "spk", "keep"))
        pe_mult = torch.from_numpy(g[pre + "pe_mult"]) if kind == "new" else None
        if step == 1:
            mid = {k: v.clone() for k, v in sd.items()}
        want_loss, want_grads = tr_ref.train_step(kind, sd, seq, spk, tgt, keep, lr, state, norm=norm, stats=stats,
                                                  pe_mult=pe_mult)
    tr = Trainer(kind, mid, lr, norm=(3.3, 2.1), stats=stats).to("cuda:0")
    loss = float(tr.step(seq, spk, tgt, keep=keep, pe_mult=pe_mult))
    assert abs(loss - float(want_loss)) <= 5e-5 * float(want_loss)
    for k, gv in tr.grads().items():
        if k in BN_FED_BIASES[kind]:
            continue
        _flip_tolerant(gv.numpy(), want_grads[k].numpy(), k)
    got = tr.state_dict()
    for k in sd:  # BatchNorm running statistics after the step (the optimiser state differs by construction)
        if k.endswith("running_var") or k.endswith("num_batches_tracked"):
            np.testing.assert_allclose(got[k].double().numpy(), sd[k].double().numpy(), rtol=1e-4, atol=1e-6, err_msg=k)


def test_training_step_matches_the_oracle_on_a_full_size_batch(gold):
    """B = 32 x L = 203 (not a multiple of 4, padding in every row): loss and every gradient against autograd on the
    CPU restatement; determinism (same step twice from the same state -> identical bits)."""
    from oracle import train_ref as tr_ref
    from dissc_amd.train import Trainer
    rs = np.random.RandomState(5)
    B, L = 32, 203
    seq = np.full((B, L), 100, dtype=np.int64)
    tgt = np.full((B, L), -1.0, dtype=np.float32)
    for b in range(B):
        n = L if b == 0 else int(rs.randint(20, L))
        seq[b, :n] = rs.randint(0, 100, size=n)
        tgt[b, :n] = rs.randint(1, 9, size=n)
    spk = rs.randint(0, 108, size=(B, 1)).astype(np.int64)
    keep = (rs.rand(B, L) <= 0.8).astype(np.float32)
    sd0 = initial_state("len")
    outs = []
    for rep in range(2):
        tr = Trainer("len", sd0, 3e-4, norm=(3.3, 2.1)).to("cuda:0")
        loss = float(tr.step(seq, spk, tgt, keep=keep))
        outs.append((loss, tr.grads(), tr.state_dict()))
    assert outs[0][0] == outs[1][0]
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k
    sd = {k: v.clone() for k, v in sd0.items()}
    want_loss, want_grads = tr_ref.train_step("len", sd, torch.from_numpy(seq), torch.from_numpy(spk), torch.from_numpy(tgt),
                                              torch.from_numpy(keep), 3e-4, {}, norm=(torch.tensor(3.3), torch.tensor(2.1)))
    assert abs(outs[0][0] - float(want_loss)) <= 5e-5 * float(want_loss)
    for k, gv in outs[0][1].items():
        if k in BN_FED_BIASES["len"]:
            continue
        _flip_tolerant(gv.numpy(), want_grads[k].numpy(), k)


def _toy_len_data(n, rs, n_spk=4):
    """a learnable rule: the run length of a unit is 1 + (unit % 3) (+1 for odd speakers)"""
    lines = []
    for i in range(n):
        spk = i % n_spk
        units = []
        for _ in range(int(rs.randint(8, 20))):
            u = int(rs.randint(0, 100))
            if units and units[-1] == u:
                continue
            units += [u] * (1 + u % 3 + (spk % 2))
        lines.append(json.dumps({"units": units, "f0": [float(100 + 10 * spk + (u % 7)) if u % 5 else 0.0 for u in units],
                                 "audio": f"s{spk}_{i:03d}.wav"}))
    return lines


def test_training_run_follows_the_oracles_loss_curve(gold):
    """48 optimisation steps on a toy rhythm corpus, the same batches and masks through the HIP engine and through the
    CPU restatement: the per-epoch training loss of the two runs stays together (measured: <0.1 % for the first epochs,
    a few % once rounding-level gradient sign flips have accumulated) and falls."""
    from oracle import train_ref as tr_ref
    from dissc_amd.train import Trainer, init_state_dict
    rs = np.random.RandomState(0)
    data = []
    for x in (json.loads(x) for x in _toy_len_data(96, rs)):
        v, l = [], []
        for u in x["units"]:
            if v and v[-1] == u:
                l[-1] += 1
            else:
                v.append(u)
                l.append(1)
        data.append((v, l, int(x["audio"][1])))
    allv = np.concatenate([d[1] for d in data]).astype(np.float32)
    norm, lr = (float(allv.mean()), float(allv.std(ddof=1))), 3e-3
    sd0 = init_state_dict("len", 100, 4, seed=1)
    tr = Trainer("len", sd0, lr, norm=norm).to("cuda:0")
    sd, st = {k: v.clone() for k, v in sd0.items()}, {}
    g = np.random.RandomState(3)
    curve = []
    for ep in range(8):
        perm, le, lo = g.permutation(96), 0.0, 0.0
        for i in range(0, 96, 16):
            idx = perm[i:i + 16]
            L = max(len(data[j][0]) for j in idx)
            seq, tgt = np.full((16, L), 100, np.int64), np.full((16, L), -1.0, np.float32)
            spk = np.zeros((16, 1), np.int64)
            for r, j in enumerate(idx):
                v, l, s = data[j]
                seq[r, :len(v)], tgt[r, :len(v)], spk[r, 0] = v, l, s
            keep = (g.rand(16, L) <= 0.8).astype(np.float32)
            le += float(tr.step(seq, spk, tgt, keep=keep))
            lo += float(tr_ref.train_step("len", sd, torch.from_numpy(seq), torch.from_numpy(spk), torch.from_numpy(tgt),
                                          torch.from_numpy(keep), lr, st,
                                          norm=(torch.tensor(norm[0]), torch.tensor(norm[1])))[0])
        curve.append((le, lo))
    for ep, (le, lo) in enumerate(curve):
        assert abs(le - lo) <= (0.002 if ep < 2 else 0.03) * lo, curve
    assert curve[-1][0] < 0.5 * curve[0][0], curve


def test_training_clis_learn_and_write_reference_checkpoints(gold, tmp_path):
    """train_len_predictor.py / train_f0_predictor.py (same flags as the reference): the loss goes down, and the
    checkpoints they write load into the inference predictors AND into plain torch modules of the reference layout."""
    td = str(tmp_path)
    rs = np.random.RandomState(0)
    os.makedirs(f"{td}/data")
    names = [f"s{i}" for i in range(4)]
    pickle.dump(names, open(f"{td}/data/id_to_spkr.pkl", "wb"))
    open(f"{td}/data/train.txt", "w").write("\n".join(_toy_len_data(96, rs)) + "\n")
    open(f"{td}/data/val.txt", "w").write("\n".join(_toy_len_data(24, rs)) + "\n")
    pickle.dump({n: {"mean": np.float64(100 + 10 * i), "std": np.float64(3.0)} for i, n in enumerate(names)},
                open(f"{td}/data/f0_stats.pkl", "wb"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_len_predictor.py"), "--out_path", f"{td}/ckpt",
                        "--data_path", f"{td}/data", "--n_epochs", "30", "--batch_size", "16", "--learning_rate", "3e-3"],
                       capture_output=True, text=True, timeout=900, cwd=td)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    log = [json.loads(x) for x in open(f"{td}/ckpt/len/log.jsonl")]
    tr_mse = [x["MSE"] for x in log if x["split"] == "train"]
    assert len(tr_mse) == 30 and tr_mse[-1] < 0.5 * tr_mse[0], tr_mse
    assert os.path.isfile(f"{td}/ckpt/len/best_model.pth") and os.path.isfile(f"{td}/ckpt/len/len_norm_stats.pth")
    # the checkpoint drives the inference predictor (what infer.py loads)
    from dissc_amd.predictors import LenPredictor
    lm = LenPredictor(100, 4).to("cuda:0")
    lm.load_state_dict(torch.load(f"{td}/ckpt/len/best_model.pth"))
    lm.norm_mean, lm.norm_std = torch.load(f"{td}/ckpt/len/len_norm_stats.pth")
    val = [x for x in log if x["split"] == "val"]
    assert val[-1]["MSE"] < 0.6 * val[0]["MSE"], val
    # unit u of speaker 0 lasts 1 + u % 3 frames: 3 4 5 7 8 30 41 17 -> 1 2 3 2 3 1 3 3
    pred = lm(torch.tensor([[3, 4, 5, 7, 8, 30, 41, 17]]), torch.tensor([[0]])).cpu().numpy()[0]
    assert np.isfinite(pred).all() and np.abs(pred - np.array([1, 2, 3, 2, 3, 1, 3, 3])).mean() < 0.7, pred
    for mt in ("new", "base"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "train_f0_predictor.py"), "--out_path", f"{td}/ckpt_{mt}",
                            "--data_path", f"{td}/data", "--f0_path", f"{td}/data/f0_stats.pkl", "--n_epochs", "10",
                            "--batch_size", "16", "--model_type", mt], capture_output=True, text=True, timeout=900, cwd=td)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        log = [json.loads(x) for x in open(f"{td}/ckpt_{mt}/pitch/log.jsonl")]
        losses = [x["loss"] for x in log if x["split"] == "train"]
        assert len(losses) == 10 and losses[-1] < 0.9 * losses[0], losses
        from dissc_amd.predictors import PitchPredictor, PitchPredictorBase
        pm = (PitchPredictorBase if mt == "base" else PitchPredictor)(100, 4).to("cuda:0")
        pm.load_state_dict(torch.load(f"{td}/ckpt_{mt}/pitch/best_model.pth"))
        out = pm.infer_freq(torch.tensor([[3, 3, 5, 5, 10]]), torch.tensor([[1]]), True)
        assert torch.isfinite(out).all()
