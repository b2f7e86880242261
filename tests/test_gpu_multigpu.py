"""The N > 1 path ON RCCL -- auto-enabled the moment a box shows two or more GPUs (every test skips on the one-GPU box, where the
same code runs over gloo on a shared device: test_gpu_cli.py, test_gpu_pipeline.py, test_gpu_rccl.py).  One command,
`pytest tests/test_gpu_multigpu.py -m gpu`, is the verdict on the first multi-GPU box:
  (a) bench.py --gpus 2 over `nccl`: two ranks seen by the process group, the collectives named, one all-gather per exchange
      round in the strong leg, the in-run parity figure still printed by rank 0;
  (b) sr/inference.py and convert.py under torch.distributed.run on RCCL write byte-identical files to the one-process run, with
      every rank writing (DISSC_WRITERS=all) and with rank 0 writing;
  (c) the committed strong-scaling model (profiles/rNN/strong_model.json) is falsifiable: measured wall within 15 % of it.
Replaces the reference's Pool(8) + integer device ids (sr/inference.py:288-292,351-354); SURVEY section 8(e)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from test_gpu_pipeline import ROOT, _free_port, _run, _sweep_jobs, _write_models, _write_wavs, models  # noqa: F401

# DISSC_MULTIGPU_REHEARSAL=gloo: run this file's code on a ONE-GPU box with two gloo ranks sharing the device (what the other test
# files do for the product path) -- a rehearsal of the TESTS themselves, so that their first execution on a multi-GPU box is not
# also the first execution of their fixtures and assertions.  Timing assertions and RCCL banners are skipped there.
REHEARSAL = os.environ.get("DISSC_MULTIGPU_REHEARSAL", "") == "gloo"
BACKEND = "gloo" if REHEARSAL else "nccl"
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2 and not REHEARSAL,
                                 reason="needs >= 2 GPUs (RCCL with one GPU per rank); DISSC_MULTIGPU_REHEARSAL=gloo rehearses on one")]


def _clean_env(**kw):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DISSC_FORCE_DIST", "DISSC_DIST_BACKEND", "DISSC_BENCH_BACKEND",
                        "DISSC_WRITERS")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if REHEARSAL:
        env.update(DISSC_BENCH_BACKEND="gloo", DISSC_DIST_BACKEND="gloo")
    env.update(kw)
    return env


def _torchrun(n, script, args):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), script] + args


@pytest.fixture(scope="module")
def bench2():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--no-split-bf16", "--no-pipeline", "--no-d2h"],
                       env=_clean_env(NCCL_DEBUG="VERSION"), capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    return json.loads(lines[0]), r.stdout + r.stderr


def test_bench_two_gpus_over_rccl(bench2):
    j, log = bench2
    assert j["n_gpus"] == 2 and j["rccl_ranks_seen"] == 2 and j["backend"] == BACKEND and j["scaling"] == "weak"
    assert BACKEND in j["config"]["collective"] and "all_gather_into_tensor" in j["config"]["collective"]
    assert j["config"]["parallelism"] == "dp2" and j["value"] > 400  # 2 x 200x real time at the very least
    assert REHEARSAL or "RCCL version" in log or "NCCL version" in log
    st = j["strong"]
    assert st["jobs"] == 1024 and len(st["per_rank_compute_ms"]) == 2 and 1.0 <= st["load_imbalance"] < 1.01
    assert 1 <= st["exchange"]["collectives"] == st["exchange"]["rounds"] <= 4  # ONE all-gather per exchange round
    assert st["own_rows"]["value"] > 400
    # rank 0's timed batch against the oracle, in the same run (no CPU timing at N > 1, the parity figure stays)
    p = j["parity"]
    assert p["utts"] >= 2 and p["rms"] <= 1e-4 and p["rel"] <= 1e-3, p
    assert "cpu_baseline" not in j


def test_strong_scaling_model_is_falsifiable_at_two_gpus(bench2):
    j, _ = bench2
    st = j["strong"]
    pred = st.get("predicted")
    if not pred or "predicted_wall_ms" not in pred:
        pytest.skip("no committed prediction for N = 2 (profiles/rNN/strong_model.json)")
    ratio = st["wall_ms"] / pred["predicted_wall_ms"]
    print(f"strong leg at 2 GPUs: measured {st['wall_ms']:.1f} ms, predicted {pred['predicted_wall_ms']:.1f} ms, ratio {ratio:.3f}")
    assert REHEARSAL or 0.85 <= ratio <= 1.15, (st["wall_ms"], pred)  # (two gloo ranks on ONE GPU say nothing about the model)


@pytest.mark.parametrize("writers", ["all", "rank0"])
def test_sr_inference_two_gpus_rccl_equals_one_process(models, golden_dir, tmp_path, writers):  # noqa: F811
    import importlib.util
    td = str(tmp_path)
    _write_models(td, models, golden_dir)
    jobs = _sweep_jobs(96, [0])
    with open(f"{td}/man.txt", "w") as f:
        for u, j in enumerate(jobs):
            f.write(json.dumps({"units": j["code"].tolist(), "f0": [float(v) for v in j["f0"]],
                                "audio": f"p{225 + u % 7}_{u:03d}.wav"}) + "\n")
    args = ["--input_code_file", f"{td}/man.txt", "--data_path", f"{td}/nowav", "--checkpoint_file", f"{td}/ckpt/",
            "--vc", "--target-speakers", "p231", "p225", "--unseen_speaker", "--id_to_spkr", f"{td}/meta/id_to_spkr.pkl", "-n", "-1"]
    script = os.path.join(ROOT, "sr", "inference.py")
    _run([sys.executable, script] + args + ["--output_dir", f"{td}/o1"], _clean_env(), td)
    r = _run(_torchrun(2, script, args + ["--output_dir", f"{td}/o2"]), _clean_env(DISSC_WRITERS=writers, NCCL_DEBUG="VERSION"), td)
    assert REHEARSAL or "RCCL version" in r.stdout + r.stderr or "NCCL version" in r.stdout + r.stderr
    files = sorted(os.listdir(f"{td}/o1"))
    assert len(files) == 192 and sorted(os.listdir(f"{td}/o2")) == files
    for fn in files:
        assert open(f"{td}/o1/{fn}", "rb").read() == open(f"{td}/o2/{fn}", "rb").read(), fn


@pytest.mark.parametrize("writers", ["all", "rank0"])
def test_convert_two_gpus_rccl_equals_one_process(models, golden_dir, tmp_path, writers):  # noqa: F811
    td = str(tmp_path)
    _write_models(td, models, golden_dir, f0_normalize=True)
    _write_wavs(f"{td}/wav", 48, seed=5, lo=2.0, hi=5.0)
    common = ["--base_dir", f"{td}/wav", "--hubert_dir", f"{td}/hub", "--len_model", f"{td}/len/", "--f0_model", f"{td}/pitch/",
              "--checkpoint_file", f"{td}/ckpt/", "--id_to_spkr", f"{td}/meta/id_to_spkr.pkl", "--target_speakers", "p231", "p225"]
    conv = os.path.join(ROOT, "convert.py")
    _run([sys.executable, conv] + common + ["--output_dir", f"{td}/a"], _clean_env(), td)
    _run(_torchrun(2, conv, common + ["--output_dir", f"{td}/b"]), _clean_env(DISSC_WRITERS=writers), td)
    files = sorted(os.listdir(f"{td}/a"))
    assert len(files) == 96 and sorted(os.listdir(f"{td}/b")) == files
    for fn in files:
        assert open(f"{td}/a/{fn}", "rb").read() == open(f"{td}/b/{fn}", "rb").read(), fn


def test_all_visible_gpus_weak_scaling_line():
    """bench.py at N = every visible GPU (what the driver's SCALE run launches): one JSON line, N ranks seen, per-GPU work fixed"""
    n = 2 if REHEARSAL else torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "2", "--no-strong"],
                       env=_clean_env(), capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == n == j["rccl_ranks_seen"] and j["config"]["batch_per_gpu"] == 32
    print(f"N = {n}: {j['value']:.0f} audio-sec/sec, {j['ms_per_step']:.2f} ms/step")
