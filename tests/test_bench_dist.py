"""bench.py's N>1 bookkeeping (barrier, per-rank inputs, all-gather shapes, max-over-ranks timing,
one JSON line on rank 0) exercised on CPU with gloo through the same launcher the driver uses.
The generator itself is replaced by a stand-in (DISSC_BENCH_FAKE=1): this checks plumbing only."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_gloo_dry_run():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, DISSC_BENCH_FAKE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--frames", "20"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["unit"] == "audio-sec/sec" and j["value"] > 0 and j["higher_is_better"] is True
    # whole-job aggregate: 2 ranks x 4 utterances x 20 frames x 20 ms per step
    assert abs(j["value"] * j["ms_per_step"] / 1e3 - 2 * 4 * 20 * 0.02) < 1e-3 * 2 * 4 * 20 * 0.02 + 1e-2
    assert "cpu_baseline" not in j and j["roofline"]["bound"] == "mfma"
    # the strong-scaling leg: the same fixed job list sharded over both ranks, ONE all-gather, per-rank times
    st = j["strong"]
    assert st["scaling"] == "strong" and st["jobs"] == 32 and st["value"] > 0
    assert len(st["per_rank_compute_ms"]) == 2 and st["compute_imbalance"] >= 1.0 and 1.0 <= st["load_imbalance"] < 1.2
    assert st["exchange"]["collectives"] == 1 and st["exchange"]["rounds"] == 1
    assert st["exchange"]["sent_bytes_per_rank"] <= 1.2 * st["exchange"]["payload_bytes_this_rank"] + 1024
    assert "all_gather_into_tensor" in j["config"]["collective"]
    assert j["rccl_ranks_seen"] == 2 and j["backend"] == "gloo"  # what the initialised process group reports


def test_bench_gpus_flag_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (how the driver runs N=1, and how a user would ask
    for N GPUs) must spawn 2 ranks itself and report n_gpus == 2 -- never time one GPU silently."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["DISSC_BENCH_FAKE"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--frames", "10"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "dp2"


def test_bench_refuses_world_size_mismatch():
    env = dict(os.environ, DISSC_BENCH_FAKE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--batch", "1", "--frames", "10"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "--gpus 2" in (r.stdout + r.stderr)


def test_cpu_baseline_and_in_run_parity_figure():
    """bench.py's cpu_baseline leg returns the oracle waveforms it produced while being timed, and parity_vs compares a
    batch against them (parity in the same run)."""
    import sys
    sys.path.insert(0, ROOT)
    import torch

    import bench
    import synthdata as synth
    sd = synth.synth_generator_state_dict(seed=0)
    code, f0, spkr, _ = synth.synth_generator_inputs(3, 12, seed=1234)
    out, waves = bench.cpu_baseline(synth, sd, torch.from_numpy(code), torch.from_numpy(f0), torch.from_numpy(spkr),
                                    budget_s=0.5, max_utts=5, node_leg=False)
    assert out["kind"] == "port" and out["value"] > 0 and len(waves) == 3 and out["node"] is None
    y = torch.stack([w.reshape(1, -1) for w in waves])
    p = bench.parity_vs(waves, y)
    assert p["rms"] == 0.0 and p["utts"] == 3 and p["tol_rms"] == 1e-4
    y[1, 0, 5] += 0.5
    p = bench.parity_vs(waves, y)
    assert p["rms"] > 1e-4 and abs(p["max"] - 0.5) < 1e-6


def test_cpu_baseline_whole_host_leg():
    """the `node` figure of cpu_baseline: P concurrent pinned B=1 oracle workers on disjoint physical cores, aggregate rate over a
    common wall-clock window (the reference's Pool(8), sr/inference.py:351-354)"""
    sys.path.insert(0, ROOT)
    import bench
    sets = bench._physical_core_sets(2)
    assert all(len(cs) == 2 for cs in sets) and len({c for cs in sets for c in cs}) == 2 * len(sets)  # disjoint
    if len(sets) < 2:
        return
    r = bench.cpu_baseline_node(2, 3, 25, duration=1.5, lead=8.0)
    if r["late_workers"]:  # a loaded machine: the workers' `import torch` outlasted the lead -- once more with a long one
        r = bench.cpu_baseline_node(2, 3, 25, duration=1.5, lead=30.0)
    assert r["workers"] == len(sets) and r["cores"] == 2 * len(sets) and r["utterances"] >= r["workers"]
    assert abs(r["value"] - r["utterances"] * 0.5 / 1.5) < 0.01 and r["late_workers"] == 0, r
