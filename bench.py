#!/usr/bin/env python
"""Headline benchmark: HiFi-GAN resynthesis throughput (audio-seconds per wall-second)
of the HIP generator on 10 s x batch-32 utterances per GPU (BASELINE.json metric,
configs[2] workload: "Batch-32 VCTK val-set resynthesis only").

    python bench.py [--gpus N --steps K --warmup W]          (N > 1: spawns N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the generator over one batch of 32 synthetic 10 s utterances
per rank (inputs resident in HBM), followed -- for N > 1 -- by the path's single RCCL
all-gather of the decoded waveforms.  Weak scaling: per-GPU work is fixed.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md (dense fp32 MFMA)
HBM_PEAK_GBPS = 8000.0         # same guide: HBM3E ~8 TB/s
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 (no sparsity)


def host_cpu_info():
    """(physical cores usable by this process, logical CPUs usable, model name) from lscpu / the affinity mask."""
    import subprocess
    avail = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    model, phys = "unknown", None
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=20).stdout
        for ln in txt.splitlines():
            if ln.startswith("Model name:"):
                model = ln.split(":", 1)[1].strip()
        # physical cores among the CPUs we may run on: distinct (socket, core) pairs
        rows = subprocess.run(["lscpu", "-p=CPU,CORE,SOCKET"], capture_output=True, text=True, timeout=20).stdout
        cores = set()
        for ln in rows.splitlines():
            if ln.startswith("#") or not ln.strip():
                continue
            cpu, core, sock = (ln.split(",") + ["0", "0"])[:3]
            if int(cpu) in avail:
                cores.add((sock, core))
        phys = len(cores) or None
    except Exception:
        pass
    return phys or len(avail), len(avail), model


PARITY_TOL_RMS = 1e-4   # north_star / BASELINE.md 4.5: RMS(gpu - cpu) <= 1e-4 ...
PARITY_TOL_REL = 1e-3   # ... and <= 1e-3 of the reference waveform's RMS
PARITY_GUARD_RMS = 2e-6  # regression guard of the exact-fp32 path (<= 10x the measured 4e-7 ... 6e-7, below split-bf16's ~4e-6;
                         # the same figure as tests/test_gpu_generator.py FP32_GUARD_RMS): beyond it the run exits 3 like a miss
                         # of the bar -- a noisier kernel must be a decision, not an accident
SCHEMA = 6              # bench line layout version (round number of the last change of workloads / fields)


def parity_vs(ref_waves, y_host):
    """RMS / relative / max error of the GPU batch `y_host` [B,1,L] against the oracle waveforms the CPU baseline leg
    produced for the first len(ref_waves) utterances of the same batch (same run, same inputs)."""
    n = len(ref_waves)
    ref = torch.stack([r.reshape(-1) for r in ref_waves]).double()
    err = y_host[:n].reshape(n, -1).double() - ref
    rms = float(err.pow(2).mean().sqrt())
    ref_rms = float(ref.pow(2).mean().sqrt())
    return {"rms": rms, "rel": rms / max(ref_rms, 1e-30), "max": float(err.abs().max()), "ref_rms": ref_rms, "utts": n,
            "worst_utt_rms": float(err.pow(2).mean(1).sqrt().max()), "tol_rms": PARITY_TOL_RMS, "tol_rel": PARITY_TOL_REL,
            "guard_rms": PARITY_GUARD_RMS, "guard": rms / PARITY_GUARD_RMS,  # measured / guard: > 1 fails the fp32 path
            "against": "oracle/generator_ref.py (CPU restatement pinned to the reference, tests/test_oracle_golden.py), B=1 per "
                       "utterance, the waveforms the cpu_baseline leg produced while being timed"}


def cpu_baseline(synth, sd, code, f0, spkr, budget_s=9.0, max_utts=400, keep=8, node_leg=True):
    """The CPU oracle (kind='port': plain-PyTorch restatement pinned to the reference,
    tests/test_oracle_golden.py) timed the way the reference runs: B=1 per utterance on the host
    cores.  Bounded sample of the same workload: thread count picked by a probe on a full 10 s
    utterance, then 3 warm-ups and >= 5 timed utterances; the value is 10 s / median time.
    `node` (BASELINE.md 4.3 "the node's own host cores"): the same stream run P = floor(physical / threads) times at once on
    disjoint cores (cpu_baseline_node) -- the reference's own deployment shape (Pool(8), sr/inference.py:351-354)."""
    from oracle import generator_ref as gr
    w = gr.fold_state_dict(sd)
    phys, logical, model = host_cpu_info()
    one = lambda b: gr.code_generator(w, synth.VCTK_CONFIG, code[b:b + 1], f0[b:b + 1], spkr[b:b + 1])
    # oneDNN degrades when oversubscribed: probe the physical and the logical count (and a few below) -- never more threads than
    # the container's CPU-time quota (the GPU boxes of this pool show 256 logical CPUs and are given 16 cores of CPU time)
    quota = cgroup_cpu_limit()
    limit = logical if quota is None else max(1, min(logical, int(quota)))
    best, threads = None, 1
    for th in sorted({min(limit, c) for c in (8, 16, 32, phys, logical)}):
        torch.set_num_threads(th)
        one(0)  # warm-up at this thread count
        t = time.perf_counter()
        one(0)
        t = time.perf_counter() - t
        if best is None or t < best:
            best, threads = t, th
    torch.set_num_threads(threads)
    for b in range(3):
        one(b % code.shape[0])
    times, t0, waves = [], time.perf_counter(), []
    while len(times) < max_utts and (len(times) < 5 or time.perf_counter() - t0 < budget_s):
        t = time.perf_counter()
        wav = one(len(times) % code.shape[0])
        times.append(time.perf_counter() - t)
        if len(waves) < min(keep, code.shape[0]):  # utterances 0 .. keep-1 of the batch, for the in-run parity figure
            waves.append(wav)
    sec = code.shape[1] * 320 / 16000.0
    med = float(np.median(times))
    node = None
    if node_leg:
        try:
            node = cpu_baseline_node(threads, code.shape[0], code.shape[1])
        except Exception as e:  # noqa: BLE001
            node = {"error": f"{type(e).__name__}: {e}"}
    return {"value": round(sec / med, 2), "unit": "audio-sec/sec", "cores": threads, "kind": "port", "node": node,
            "physical_cores": phys, "logical_cpus": logical, "cpu_model": model, "cgroup_cpu_quota_cores": quota,
            "sample": f"{len(times)} x {sec:g} s utterances, B=1 each (reference style), torch CPU fp32, {threads} threads "
                      f"(best of a probe on a full utterance), 3 warm-ups, median of {len(times)} "
                      f"(mean rate {sec * len(times) / sum(times):.2f}), {sum(times):.1f} s of CPU work"}, waves


def cgroup_cpu_limit():
    """CPU-time quota of this container in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unreadable.  A box
    can show 256 logical CPUs in its affinity mask and still be throttled to a fraction of them."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def _physical_core_sets(threads):
    """Disjoint sets of `threads` logical CPUs, ONE per physical core, neighbours in (socket, core) order -- the cores this
    process may run on, cut into floor(physical / threads) groups."""
    import subprocess
    avail = set(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else set(range(os.cpu_count() or 1))
    first = {}
    try:
        rows = subprocess.run(["lscpu", "-p=CPU,CORE,SOCKET"], capture_output=True, text=True, timeout=20).stdout
        for ln in rows.splitlines():
            if ln.startswith("#") or not ln.strip():
                continue
            cpu, core, sock = (ln.split(",") + ["0", "0"])[:3]
            if int(cpu) in avail:
                first.setdefault((int(sock), int(core)), int(cpu))
    except Exception:
        pass
    cpus = [first[k] for k in sorted(first)] or sorted(avail)
    return [cpus[i:i + threads] for i in range(0, len(cpus) - threads + 1, threads)]


def cpu_node_worker(argv):
    """`python bench.py --cpu-node-worker T0 DURATION THREADS B T cpu,cpu,...`: one B=1 oracle stream pinned to its own cores
    (a worker of cpu_baseline's whole-host leg; the reference's deployment is a Pool(8) of such workers, sr/inference.py:351-354).
    Prints one JSON line: utterances finished inside [T0, T0 + DURATION)."""
    t0, dur, threads, B, T = float(argv[0]), float(argv[1]), int(argv[2]), int(argv[3]), int(argv[4])
    cpus = [int(c) for c in argv[5].split(",")]
    if hasattr(os, "sched_setaffinity"):
        os.sched_setaffinity(0, cpus)
    torch.set_num_threads(threads)
    import synthdata as synth
    from oracle import generator_ref as gr
    w = gr.fold_state_dict(synth.synth_generator_state_dict(seed=0))
    code, f0, spkr, _ = synth.synth_generator_inputs(B, T, seed=1234)
    code, f0, spkr = torch.from_numpy(code), torch.from_numpy(f0), torch.from_numpy(spkr)
    one = lambda b: gr.code_generator(w, synth.VCTK_CONFIG, code[b:b + 1], f0[b:b + 1], spkr[b:b + 1])
    one(0)
    one(1 % B)
    ready = time.time()
    while time.time() < t0:
        time.sleep(0.001)
    n, ends = 0, []
    while time.time() < t0 + dur:
        one(n % B)
        n += 1
        ends.append(time.time() - t0)
    done = sum(1 for e in ends if e <= dur)
    print(json.dumps({"done": done, "late_start_s": max(0.0, ready - t0), "last_end": ends[-1] if ends else 0.0}), flush=True)


def cpu_baseline_node(threads, B, T, duration=5.0, lead=12.0):
    """Whole-host figure next to the single-stream one: P = floor(physical cores / threads) concurrent B=1 oracle workers on
    disjoint core sets (fresh processes, `threads` threads each, same utterances), all measuring the same `duration` seconds of
    wall time; aggregate audio-sec/sec = finished utterances x utterance length / duration."""
    import subprocess
    sets = _physical_core_sets(threads)
    quota = cgroup_cpu_limit()
    if quota is not None:  # never start more busy threads than the container is given CPU time for
        sets = sets[:max(0, int(quota) // threads)]
    if len(sets) < 2:
        return {"skipped": f"{len(sets)} core set(s) of {threads} threads on this host"
                           + (f" (cgroup CPU quota {quota:g} cores)" if quota is not None else "")}
    t0 = time.time() + lead  # the workers import torch, fold the weights and warm up before T0
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-node-worker", repr(t0), str(duration), str(threads),
                               str(B), str(T), ",".join(map(str, cs))], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                              env=dict(os.environ, OMP_NUM_THREADS=str(threads), CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
             for cs in sets]
    res = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=lead + duration + 60)
            res.append(json.loads(out.strip().splitlines()[-1]))
        except Exception as e:  # noqa: BLE001
            pr.kill()
            res.append({"done": 0, "error": f"{type(e).__name__}: {e}"})
    sec = T * 320 / 16000.0
    done = sum(r.get("done", 0) for r in res)
    return {"value": round(done * sec / duration, 2), "unit": "audio-sec/sec", "workers": len(sets), "threads_per_worker": threads,
            "cores": len(sets) * threads, "utterances": done, "window_s": duration,
            "late_workers": sum(1 for r in res if r.get("late_start_s", 0) > 0 or "error" in r), "cgroup_cpu_quota_cores": quota,
            "per_worker_utterances": [r.get("done", 0) for r in res],
            "sample": f"{len(sets)} concurrent B=1 oracle processes x {threads} threads on disjoint physical cores "
                      f"(sched_setaffinity), {done} x {sec:g} s utterances finished in a common {duration:g} s window"}


class _FakeGenerator:
    """CPU stand-in used only by the DISSC_BENCH_FAKE dry run."""

    h = {"upsample_rates": [5, 4, 4, 2, 2]}

    def __call__(self, code, f0, spkr, lengths=None):
        return (code.float().mean(1, keepdim=True) + spkr.float()).unsqueeze(2).expand(-1, 1, 320 * code.shape[1]).contiguous()

    def flops(self, frames):
        return 321.664e6 * frames

    flops_executed = flops


class _FakeEvent:
    def __init__(self, enable_timing=True):
        self.t = 0.0

    def record(self):
        self.t = time.perf_counter()

    def synchronize(self):
        pass

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def calibrate_len_model(lm, units, frames, spk, iters=4):
    """Give the SYNTHETIC length predictor (random weights) statistics under which a conversion roughly preserves the
    duration: sum of predicted frames ~= source frames, as a trained rhythm model does on average (reference
    infer.py:24-45 keeps the utterance's units and re-times them).  Without this the synthetic checkpoint turns 320 s of
    input into ~72 s of output and the generator stage does a quarter of a real conversion's work.  Untimed set-up:
    the raw (de-normalised with mean 0 / std 1) outputs fix the spread, a few fixed-point steps fix the mean against
    the clamp-at-one-frame rule of the rounding stage."""
    from dissc_amd import predictors as P
    vals, _, n = P.dedup(units, frames)
    lm.norm_mean, lm.norm_std = torch.tensor(0.0), torch.tensor(1.0)
    raw = lm(vals, spk, lengths=n)
    valid = torch.arange(raw.shape[1], device=raw.device)[None, :] < n[:, None]
    r = raw[valid].double()
    mu, sd = float(r.mean()), float(r.std())
    want = float(frames.sum()) / float(n.sum())      # mean frames per dedup'd unit of the source
    target = want
    for _ in range(iters):
        std = 0.35 * want / max(sd, 1e-6)
        lm.norm_mean, lm.norm_std = torch.tensor(target - std * mu, dtype=torch.float32), torch.tensor(std, dtype=torch.float32)
        tot = P.infer_batch(units, frames, spk, lm, None)["totals"]
        ratio = float(tot.sum()) / float(frames.sum())
        target += want * (1.0 - ratio)
    return ratio


def pipeline_leg(synth, dev, g, utts=32, seconds=10.0, iters=5):
    """"also report the full pipeline separately" -- encode -> len / pitch prediction -> resynthesis of
    `utts` x `seconds` s of synthetic audio through the device-resident Converter (dissc_amd/pipeline.py), wall time of
    the whole call with the input audio already in HBM (like `value`), one converted utterance per input.  The
    synthetic rhythm model is calibrated to preserve the duration (output_audio_sec ~= input), and a separate profiled
    pass splits the batch time into stages with HIP events."""
    import time

    import numpy as np
    import torch
    from dissc_amd import predictors as P
    from dissc_amd.hubert import HubertEncoder
    from dissc_amd.pipeline import Converter
    n = int(seconds * 16000)
    hsd = synth.synth_hubert_state_dict(6)
    enc = HubertEncoder(hsd, synth.synth_kmeans_centers(), 6).to(dev)
    lm = P.LenPredictor(100, 108).to(dev)
    lm.load_state_dict(synth.synth_len_state_dict(100, 108))
    # the pitch model type the reference's conversion scripts use for the VCTK / ESD checkpoints
    # (scripts/convert_eval.py:82-85); it has no 850-frame positional-encoding limit
    pm = P.PitchPredictorBase(100, 108).to(dev)
    pm.load_state_dict(synth.synth_pitch_state_dict("base", 100, 108))
    waves = [torch.from_numpy(synth.synth_waveform(n, seed=i)).to(dev) for i in range(utts)]
    ns = torch.full((utts,), n, dtype=torch.int32)
    # Untimed set-up of a REPRESENTATIVE synthetic unit stream: random centroids against a random-weight HuBERT give
    # degenerate units (whole utterances of one unit), so the codebook is re-drawn from the encoder's own features
    # (100 seeded frames of these utterances): units then change every few frames like k-means units of speech do.
    e = enc(torch.stack(waves), n_samples=ns, want_dense=True)
    dense = e["dense"].reshape(-1, e["dense"].shape[-1])
    pick = torch.from_numpy(np.random.RandomState(17).choice(dense.shape[0], 100, replace=False)).to(dev)
    enc = HubertEncoder(hsd, dense.index_select(0, pick).float().cpu(), 6).to(dev)
    e = enc(torch.stack(waves), n_samples=ns, want_dense=False)
    spk = torch.full((utts, 1), 6, dtype=torch.int64)
    ratio = calibrate_len_model(lm, e["units"], e["frames"].to(dev), spk)
    _, _, n_units = P.dedup(e["units"], e["frames"].to(dev))
    mean_run = float(e["frames"].sum()) / max(float(n_units.sum()), 1.0)
    conv = Converter(enc, lm, pm, g)
    for _ in range(3):  # steady state: the untimed set-up above leaves the chip idle for a while (cold runs read slow)
        out = conv(waves, [6])
    out_sec = sum(len(w) for w in out.values()) / 16000.0
    ts = []
    for _ in range(iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = conv(waves, [6])
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    wall = float(np.median(ts))
    # stage split: HIP events between the stages of separate, profiled calls (median per stage)
    profs = []
    for _ in range(iters):
        conv.profile = {}
        conv(waves, [6])
        profs.append(conv.profile)
    conv.profile = None
    st = {k: float(np.median([p.get(k, 0.0) for p in profs])) for k in profs[0]}
    stages = {"encode_ms": round(st.get("encode_ms", 0.0), 2), "predict_ms": round(st.get("predict_ms", 0.0), 2),
              "generator_ms": round(st.get("generator_ms", 0.0), 2),
              "host_ms": round(st.get("stage_ms", 0.0) + st.get("pack_d2h_ms", 0.0), 2)}
    return {"workload": f"encode (HuBERT-6L + k-means) -> length / pitch predictors -> HiFi-GAN, {utts} x {seconds:g} s in, "
                        "1 target speaker, base pitch model, in memory (dissc_amd.pipeline.Converter), input audio resident in "
                        "HBM; synthetic codebook drawn from the encoder's own features and synthetic rhythm model calibrated "
                        "to preserve the duration (untimed set-up)",
            "ms_per_batch": round(wall * 1e3, 2), "value": round(utts * seconds / wall, 1),
            "unit": "input audio-sec/sec", "input_audio_sec": utts * seconds, "output_audio_sec": round(out_sec, 1),
            "duration_ratio": round(ratio, 3), "mean_frames_per_unit": round(mean_run, 2), "iters": iters,
            "stages": dict(stages, sum_ms=round(sum(stages.values()), 2),
                           note="HIP-event spans of a profiled pass; host_ms = padding of the batch on the device + "
                                "ragged pack + D2H of the waveforms + unpack")}


def d2h_leg(g, d_code, d_f0, d_spkr, steps, audio_sec_per_step):
    """the same step INCLUDING the GPU post-processing and the device-to-host copy of the decoded
    waveforms (page-locked destination); a second figure, never `value`."""
    from dissc_amd.generator import wav_postprocess_
    B, L = d_code.shape[0], d_code.shape[1] * 320
    n = torch.full((B,), L, dtype=torch.int32, device=d_code.device)
    host = torch.empty(B, 1, L, dtype=torch.float32, pin_memory=True)
    for _ in range(2):
        y = g(code=d_code, f0=d_f0, spkr=d_spkr)
        wav_postprocess_(y, n)
        host.copy_(y, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        y = g(code=d_code, f0=d_f0, spkr=d_spkr)
        wav_postprocess_(y, n)
        host.copy_(y, non_blocking=True)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"what": "generator + int16-truncate / peak-normalise on the GPU + D2H of the waveforms into page-locked memory, "
                    "host-synchronised every step", "ms_per_step": round(dt * 1e3, 3),
            "value": round(audio_sec_per_step / dt, 1), "unit": "audio-sec/sec"}


def strong_jobs(synth, n_utts=256, targets=(6, 57, 3, 101), seed=77):
    """The fixed job list of the strong-scaling figure: 256 ragged utterances of 2-5 s (100-250
    frames, BASELINE configs[3]/[4] shaped) x 4 target speakers = 1 024 generator jobs, the same list at every N."""
    rs = np.random.RandomState(seed)
    jobs = []
    for u in range(n_utts):
        T = int(rs.randint(100, 251))
        code, f0, _, _ = synth.synth_generator_inputs(1, T, seed=7000 + u)
        for t in targets:
            jobs.append(dict(code=code[0], f0=f0[0, 0], spkr=int(t)))
    return jobs


XGMI_RING_GBPS = 60.0  # conservative effective all-gather bandwidth per rank over xGMI (one ring direction of a
#                        ~76.8 GB/s-per-direction link; RCCL normally drives several rings) -- only used by the model below


def strong_model(n):
    """The committed strong-scaling prediction for N ranks (profiles/rNN/strong_model.json, written by
    tools/strong_rehearsal.py from one-GPU rehearsals: every rank's share computed ALONE on the GPU + rank 0's measured
    delivery cost + a modelled all-gather), or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "strong_model.json")), reverse=True):
        try:
            j = json.load(open(path))
            row = j["per_n"].get(str(n))
            if row:
                return dict(row, source=os.path.relpath(path, ROOT))
        except Exception:  # noqa: BLE001
            continue
    return None


def strong_leg(synth, g, dev, rank, world, dist, fake=False, reps=3):
    """Strong scaling: the fixed 1 024-job list LPT-sharded over the ranks by length, each rank batching its share
    through the generator + GPU post-processing in rounds, ONE all-gather of the ragged exchange buffer per round, rank
    0 receiving every waveform on the host -- round k's gather / device-to-host copy / hand-over to the sink running on
    the harness's delivery thread while round k + 1 computes.  Wall = barrier -> slowest rank done (rank 0's delivery
    of the last round included).  Next to it: each rank's own GPU span, when its last kernel finished, rank 0's
    delivery split, the exposed tail, and the committed prediction for this N (strong_model).
    DISSC_STRONG_EMULATE="1,2,4,8" (one process, N = 1): additionally computes every rank's share of an N-rank run
    alone, for tools/strong_rehearsal.py's prediction."""
    from dissc_amd import harness
    post = None
    if not fake:
        from dissc_amd.generator import wav_postprocess_ as post
    jobs = strong_jobs(synth) if not fake else strong_jobs(synth, 16, (6, 57))
    lengths = [len(j["code"]) for j in jobs]
    audio_sec = sum(lengths) * 320 / 16000.0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(own_rows):
        best = None
        for rep in range(reps + 1):  # first pass = warm-up (workspace, staging buffers)
            barrier()
            stats = {}
            t0 = time.perf_counter()
            got = {}

            def sink(waves):  # what sr/inference.py's sink gets per round (it writes the files): host arrays, valid during the call
                for j, w in waves.items():
                    got[j] = len(w)
            harness.run_resynthesis(g, jobs, rank, world, dev, dist, postprocess=post, stats=stats, sink=sink,
                                    own_rows=own_rows)
            torch.cuda.synchronize()
            t_rank = time.perf_counter() - t0
            if dist is not None:
                dist.barrier()
            wall = time.perf_counter() - t0
            mine = set(harness.lpt_shard(lengths, world)[rank]) if own_rows else (set(range(len(jobs))) if rank == 0 else set())
            assert set(got) == mine and all(got[j] == 320 * lengths[j] for j in got)
            rec = torch.tensor([wall, stats["compute_s"], t_rank, stats["compute_done_s"], stats.get("gather_wait_s", 0.0),
                                stats.get("unpack_s", 0.0), stats.get("sink_s", 0.0), stats.get("pack_s", 0.0),
                                stats.get("submit_wait_s", 0.0), stats["host_batching_s"]], dtype=torch.float64, device=dev)
            if dist is not None:
                allr = torch.empty(world, rec.numel(), dtype=torch.float64, device=dev)
                dist.all_gather_into_tensor(allr, rec[None])
            else:
                allr = rec[None]
            allr = allr.cpu().numpy()
            cur = (float(allr[:, 0].max()), allr, stats)
            if rep > 0 and (best is None or cur[0] < best[0]):
                best = cur
        return best

    wall, allr, stats = run(False)
    comp = allr[:, 1]
    ms = lambda v: round(float(v) * 1e3, 2)
    out = {"workload": f"{len(jobs)} generator jobs ({len(jobs) // 4 if not fake else len(jobs) // 2} ragged utterances of 2-5 s x "
                       f"{4 if not fake else 2} targets), the same list at every N, LPT-sharded by length; host batching + H2D + "
                       "generator + GPU post-processing + ragged pack + all-gather + rank-0 D2H into page-locked memory + hand-over "
                       "to a per-round sink all inside the wall; rounds delivered by a worker thread while the next one computes",
           "scaling": "strong", "jobs": len(jobs), "audio_sec": round(audio_sec, 1),
           "wall_ms": ms(wall), "value": round(audio_sec / wall, 1), "unit": "audio-sec/sec",
           "per_rank_compute_ms": [ms(c) for c in comp],
           "per_rank_compute_done_ms": [ms(c) for c in allr[:, 3]],
           "exposed_tail_ms": ms(wall - allr[:, 3].max()),
           "rank0_delivery": {"wait_for_round_ms": ms(allr[0, 4]), "d2h_unpack_ms": ms(allr[0, 5]), "sink_ms": ms(allr[0, 6]),
                              "pack_ms": ms(allr[0, 7]), "blocked_on_delivery_ms": ms(allr[0, 8]),
                              "host_batching_ms": ms(allr[0, 9]),
                              "note": "delivery thread of rank 0, summed over the rounds; wait_for_round includes waiting for the "
                                      "round's compute (overlapped), blocked_on_delivery = main thread waiting for a free slot"},
           "compute_imbalance": round(float(comp.max() / comp.mean()), 4),
           "load_imbalance": round(float(stats["imbalance"]), 4),
           "exchange": {"collectives": int(stats.get("collectives", 0)), "rounds": int(stats["rounds"]),
                        "overlap": bool(stats.get("overlap", False)),
                        "sent_bytes_per_rank": 4 * int(stats["sent_floats"]),
                        "payload_bytes_this_rank": 4 * int(stats["payload_floats"])},
           "best_of": reps}
    if world > 1:  # the CLIs' default at N > 1: every rank drains and "writes" the rows it decoded itself
        w2, a2, s2 = run(True)
        out["own_rows"] = {"what": "same list, every rank delivers the rows it decoded (DISSC_WRITERS=all, the CLIs' default at N > 1); "
                                   "the round's collective carries the row tables only (DISSC_OWN_ROWS_GATHER=full: whole buffers)",
                           "wall_ms": ms(w2), "value": round(audio_sec / w2, 1),
                           "exposed_tail_ms": ms(w2 - a2[:, 3].max()),
                           "exchange": {"collectives": int(s2.get("collectives", 0)), "rounds": int(s2["rounds"]),
                                        "sent_bytes_per_rank": 4 * int(s2["sent_floats"]),
                                        "rows_all_ranks": int(s2.get("rows_all_ranks", 0))}}
    emu = os.environ.get("DISSC_STRONG_EMULATE", "")
    if emu and world == 1 and not fake:
        # Every rank's share at N ranks (the LPT partition, the rounds and the batches of the N-rank run) computed in THIS
        # process, one share after the other, the GPU to itself.  (Rehearsing with N processes on one GPU does not give
        # these numbers: beyond two processes the hardware queues are oversubscribed and every share takes 2-4x as long
        # whatever its size -- measured, profiles/r04/README.md.)
        out["emulated"] = {}
        for n in [int(v) for v in emu.split(",") if v.strip()]:
            parts = harness.lpt_shard(lengths, n)
            bud = harness.overlap_budget(lengths, parts, harness.ROUND_FLOATS // 320, cap=harness.OVERLAP_CAP_FLOATS // 320)   # the round cut of the N-rank run
            rf = None if bud is None else ([b * 320 for b in bud] if isinstance(bud, list) else bud * 320)
            walls, spans, rounds = [], [], 0
            for r in range(n):
                sub = [jobs[i] for i in parts[r]]
                ts = []
                for rep in range(3):
                    torch.cuda.synchronize()
                    st = {}
                    t0 = time.perf_counter()
                    harness.run_resynthesis(g, sub, 0, 1, dev, None, postprocess=post, stats=st, unpack_ranks=(),
                                            round_floats=rf, overlap=False)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0, st["compute_s"], st["rounds"]))
                walls.append(min(t[0] for t in ts))
                spans.append(min(t[1] for t in ts))
                rounds = max(rounds, ts[0][2])
            full = harness.plan_rounds(lengths, parts, bud)
            n_cap_dc = [harness.pack_geometry(lengths, sh, 320) for sh in full]
            out["emulated"][str(n)] = {
                "share_wall_ms": [ms(v) for v in walls], "share_gpu_span_ms": [ms(v) for v in spans], "rounds": len(full),
                "sent_bytes_per_rank": 4 * sum(harness.buffer_floats(a, b) for a, b in n_cap_dc),
                "round_fractions": [round(sum(lengths[i] for p in sh for i in p) / float(sum(lengths)), 4) for sh in full],
                "load_imbalance": round(float(harness.imbalance(lengths, parts)), 4)}
        out["emulated"]["what"] = ("each rank's share at N ranks (same partition, rounds and batches) computed alone in one "
                                   "process: host batching + H2D + generator + post-processing + ragged pack, no exchange")
    pred = strong_model(world)
    if pred is not None:
        out["predicted"] = pred
        # the committed prediction for this N next to what was just measured (round 5 verdict, item 6b): with N > 1 on real
        # hardware this is the first check of the model; relative error = (measured - predicted) / predicted
        if isinstance(out.get("wall_ms"), (int, float)) and pred.get("predicted_wall_ms"):
            out["vs_predicted"] = {"wall_ms": out["wall_ms"], "predicted_wall_ms": pred["predicted_wall_ms"],
                                   "rel_err": round((out["wall_ms"] - pred["predicted_wall_ms"]) / pred["predicted_wall_ms"], 4),
                                   "source": pred.get("source")}
    return out


def latency_leg(synth, g, dev, frames=(100, 500), reps=50):
    """The reference's own operating point (B = 1, one utterance at a time: sr/inference.py:67-76,178; infer.py:24-45): latency of
    one generator forward, T in `frames`, the median of `reps` forwards timed one by one (HIP events) after the main loop has warmed
    the chip up -- 20-launch cold runs read 10 % slow.  Small grids step down to smaller tiles with the same bits (conv32_pick_cfg,
    launch_wino_t, run_wino8: DISSC_W8_SMALL), so these waveforms are bitwise the B = 32 ones."""
    out = {}
    for T in frames:
        code, f0, spkr, _ = synth.synth_generator_inputs(1, T, seed=1234)
        c, f, s = (torch.from_numpy(v).to(dev) for v in (code, f0, spkr))
        for _ in range(10):
            g(code=c, f0=f, spkr=s)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g(code=c, f0=f, spkr=s)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        med = ts[len(ts) // 2]
        out[f"T{T}"] = {"ms": round(med, 3), "p10_ms": round(ts[len(ts) // 10], 3), "p90_ms": round(ts[(9 * len(ts)) // 10], 3),
                        "x_realtime": round(T * 0.02 / med * 1e3, 1), "tflops_algorithmic": round(g.flops(T) / med / 1e9, 1)}
    out["note"] = f"B = 1, median of {reps} forwards timed one at a time (launch + kernels, inputs resident), exact fp32"
    return out


def split_bf16_leg(synth, sd, dev, d_code, d_f0, d_spkr, y_fp32, steps, audio_sec_per_step, flops_step):
    """The opt-in split-bf16 ("bf16x3") arithmetic mode of the same generator, timed on the same
    batch right after the fp32 run and checked against the fp32 waveform (north_star bar: 1e-4 RMS).
    Reported next to the headline number, never instead of it."""
    import dissc_amd
    g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG, precision="split_bf16").to(dev)
    g.load_state_dict(sd)
    g.eval()
    g.remove_weight_norm()
    for _ in range(2):
        y = g(code=d_code, f0=d_f0, spkr=d_spkr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        y = g(code=d_code, f0=d_f0, spkr=d_spkr)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    err = (y - y_fp32).double()
    return {"arithmetic": "bf16 hi/lo operand split, 3 bf16 MFMAs per product, fp32 accumulate "
                          "(CodeGenerator(h, precision='split_bf16') / dissc_set_option('precision', 1); default is exact fp32)",
            "ms_per_step": round(dt * 1e3, 3), "value": round(audio_sec_per_step / dt, 1),
            "unit": "audio-sec/sec", "_y_head": y[:8].cpu(),
            # 3 bf16 MFMA products per algorithmic multiply-add, against the dense bf16 peak
            "roofline": {"bound": "mfma", "achieved": round(3 * flops_step / dt / 1e12, 1), "peak": BF16_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s (bf16)", "frac": round(3 * flops_step / dt / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4)},
            "rms_vs_fp32": float(err.pow(2).mean().sqrt()),
            "max_abs_vs_fp32": float(err.abs().max()), "tolerance_rms": 1e-4}


GEN_KERNEL_SOURCES = ("common.h", "conv_epilogue32.h", "conv_host.hip", "conv_mfma.hip", "conv_mfma32.hip", "conv_wino.hip",
                      "conv_wino8.hip", "gen_misc.hip", "generator.hip", "pair_host.hip", "respair.hip", "respair16_f23.hip", "respair_f23.h", "respair_f23.hip",
                      "wino_common.h")


def kernel_source_hash():
    """sha256 over the sources of the generator's default-path kernels: what a PMC capture is valid for."""
    import hashlib
    h = hashlib.sha256()
    for fn in GEN_KERNEL_SOURCES:
        with open(os.path.join(ROOT, "dissc_amd", "csrc", fn), "rb") as f:
            h.update(fn.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def hbm_traffic(B, T):
    """HBM bytes per step from the latest committed PMC capture of these kernels (profiles/rNN/hbm_traffic.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, gfx950 correction applied,
    tools/capture_profiles.sh + tools/prof_tables.py); only valid for the shape it was captured on AND for the
    kernel sources it was captured from: the capture stores `kernel_source_hash`, and a capture of other sources
    is refused (traffic null, the reason in `traffic_source`).  PMC counters cannot be read from inside the
    timed run, so this is a committed measurement, named in the output."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "hbm_traffic.json")), reverse=True):
        try:
            j = json.load(open(path))
        except Exception:
            continue
        rel = os.path.relpath(path, ROOT)
        if (B, T) != (32, 500):
            return None, f"{rel}: captured at B=32 x T=500 only"
        if j.get("kernel_source_hash") != kernel_source_hash():
            return None, (f"{rel}: STALE -- captured from kernel sources {j.get('kernel_source_hash')}, "
                          f"this build is {kernel_source_hash()} (re-run tools/capture_profiles.sh)")
        return j["bytes_per_step_B32_T500"], rel + ": " + j.get("kernel_version", "")
    return None, None


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-node-worker":
        return cpu_node_worker(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-split-bf16", action="store_true", help="skip the extra split-bf16 leg (N=1 only)")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the full-pipeline leg (N=1 only)")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling leg (fixed 1 024-job list)")
    ap.add_argument("--no-d2h", action="store_true", help="skip the D2H-inclusive figure (N=1 only)")
    ap.add_argument("--no-latency", action="store_true", help="skip the B = 1 latency leg (N=1 only)")
    ap.add_argument("--strong-only", action="store_true",
                    help="only the strong-scaling leg (tools/strong_rehearsal.py): prints {'strong': ...} and exits")
    a = ap.parse_args()

    # DISSC_BENCH_FAKE=1: CPU/gloo dry run of the distributed bookkeeping only (tests/test_bench_dist.py);
    # it measures nothing and never replaces the HIP path in a real run.
    fake = os.environ.get("DISSC_BENCH_FAKE") == "1"
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: re-exec this same command under
        # torch.distributed.run, one rank per GPU (the launch line the driver itself uses for N > 1).
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    dist = None
    backend = "gloo" if fake else os.environ.get("DISSC_BENCH_BACKEND", "nccl")
    # DISSC_FORCE_DIST=1: initialise the process group with ONE rank too, so that the collectives of the path run
    # on RCCL on a one-GPU box (the default N=1 line has no collective and no process group)
    if world > 1 or os.environ.get("DISSC_FORCE_DIST", "0") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if fake:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            # DISSC_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks
            # (ranks share devices round-robin, gloo stages the device tensors through the host);
            # the driver's runs use the default, RCCL with one GPU per rank.
            if backend != "nccl":
                local_rank = local_rank % torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
            kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    n_gpus = world
    if n_gpus != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus N spawns them itself)")
    if not fake and world > 1 and backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks need {world} GPUs, {torch.cuda.device_count()} visible")
    import synthdata as synth  # deterministic synthetic checkpoints / inputs
    sd = synth.synth_generator_state_dict(seed=0)
    if fake:
        dev = torch.device("cpu")
        g = _FakeGenerator()
        torch.cuda.synchronize = lambda *a, **k: None
        torch.cuda.Event = _FakeEvent
    else:
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)
        import dissc_amd
        g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to(dev)
        g.load_state_dict(sd)
        g.eval()
        g.remove_weight_norm()

    if not fake and world > 1 and backend == "nccl":
        from dissc_amd import harness as _h
        _h.pin_to_gpu_numa(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))  # host threads next to the GPU
    if a.strong_only:
        strong = strong_leg(synth, g, dev, rank, world, dist, fake)
        if rank == 0:
            print(json.dumps({"strong": strong, "n_gpus": world, "backend": backend if dist is not None else None}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    B, T = a.batch, a.frames
    code, f0, spkr, _ = synth.synth_generator_inputs(B, T, seed=1234 + rank)
    d_code = torch.from_numpy(code).to(dev)
    d_f0 = torch.from_numpy(f0).to(dev)
    d_spkr = torch.from_numpy(spkr).to(dev)
    hop = 320
    gathered = torch.empty(world * B, 1, hop * T, device=dev) if dist is not None else None

    def step():
        y = g(code=d_code, f0=d_f0, spkr=d_spkr)
        if dist is not None:
            dist.all_gather_into_tensor(gathered, y)  # the path's single RCCL collective
        return y

    for _ in range(a.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gen_ms = 0.0
    pending = None  # (work, y): the previous step's all-gather, still in flight on RCCL's stream
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ev0.record()
        y = g(code=d_code, f0=d_f0, spkr=d_spkr)
        ev1.record()
        if dist is not None:
            # The gather of step k runs on RCCL's stream while step k+1's generator kernels run
            # on the compute stream; it is only waited for right before the next gather reuses
            # `gathered` (and after the loop), so all K gathers are inside the timed region.
            if pending is not None:
                pending[0].wait()
            pending = (dist.all_gather_into_tensor(gathered, y, async_op=True), y)
        ev1.synchronize()
        gen_ms += ev0.elapsed_time(ev1)
    if pending is not None:
        pending[0].wait()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    audio_sec_per_step = world * B * T * hop / 16000.0
    value = audio_sec_per_step * a.steps / dt
    y_head = y[:8].cpu() if rank == 0 else None  # the timed loop's last fp32 batch, for the in-run parity figure

    # strong-scaling figure (every rank takes part; rank 0 reports): a fixed ragged job list at every N
    strong = None
    if not a.no_strong:
        try:
            strong = strong_leg(synth, g, dev, rank, world, dist, fake)
        except Exception as e:  # noqa: BLE001
            if dist is not None:
                raise  # a rank that drops out of a collective must not leave the others waiting
            strong = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        flops_step = g.flops(B * T)          # algorithmic 2*MAC per rank per step (direct form)
        flops_exec = g.flops_executed(B * T)  # what the matrix pipe executes (Toom-Cook layers do fewer products)
        kern_s = gen_ms / a.steps / 1e3      # HIP-event time of the generator launches
        ach = flops_exec / kern_s / 1e12      # the roofline fraction is taken on EXECUTED work, so it stays <= 1
        ach_alg = flops_step / kern_s / 1e12
        out = {
            "schema": SCHEMA,
            "metric": "audio-sec/sec (RTF) HiFi-GAN resynth, 10s x32 batch",
            "value": round(value, 1), "unit": "audio-sec/sec", "n_gpus": n_gpus,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "backend": backend if dist is not None else None,
            "rccl_ranks_seen": (dist.get_world_size() if dist is not None else 1),  # what the initialised process group reports
            "data": "synthetic (seeded codes/f0/speakers, seeded random weights in the reference checkpoint layout)",
            "config": {"workload": "HiFi-GAN generator only (sr/inference.py generate()), "
                                   f"B={B} x T={T} frames (10 s @16 kHz) per GPU, VCTK hubert100_lut config",
                       "batch_per_gpu": B, "frames": T, "parallelism": f"dp{n_gpus}",
                       "collective": (f"1 all_gather_into_tensor ({backend}) of the step's waveforms [B,1,L] per rank and step, "
                                      "asynchronous, overlapping the next step's kernels; `strong` leg: 1 all_gather_into_tensor of "
                                      "the ragged exchange buffer per run; bookkeeping outside the timed regions: barrier, "
                                      "all_reduce(MAX) of the wall time, all_gather of 3 timing doubles")
                       if dist is not None else "none (no process group at N=1)"},
            "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                         "traffic": hbm_traffic(B, T)[0], "traffic_source": hbm_traffic(B, T)[1],
                         "kernel": "all generator convs: conv_wino_kernel / conv_wino8_kernel (Toom-Cook F(4,3) on 12-wave workgroups, F(6,3) and "
                                   "F(5,4) on 8-wave workgroups, over 3- or 4-tap sub-filters, fp32 v_mfma_f32_32x32x2) on the ResBlocks of the C >= 64 stages, "
                                   "conv_mfma32_kernel (direct, same MFMA) for conv_pre and the ConvTranspose layers, respair32/respair16 "
                                   "fused residual pairs on the C = 32 / 16 stages (direct; their k = 11 pairs as register-only F(2,3): respair32_f23_kernel / respair16_f23_kernel)",
                         "algorithmic_tflops": round(ach_alg, 2),
                         "algorithmic_frac": round(ach_alg / FP32_MFMA_PEAK_TFLOPS, 4),
                         "note": "achieved / frac = EXECUTED matrix-pipe FLOPs / time (<= peak by construction; the Toom-Cook layers "
                                 "execute 6 ceil(k/3) / 4 -- F(4,3) --, 8 ceil(k/3) / 6 -- F(6,3) -- or 8 ceil(k/4) / 5 -- F(5,4) -- products per output instead of k); algorithmic_tflops / algorithmic_frac = "
                                 "direct-form FLOPs (SURVEY 8d: 321.664 MFLOP per frame) / time -- an effective rate, not pipe utilisation",
                         "flops_per_step": flops_step, "flops_executed": flops_exec,
                         "kernel_ms_per_step": round(kern_s * 1e3, 3)},
        }
        traffic = out["roofline"]["traffic"]
        if traffic:  # north_star also asks for the fraction of the HBM roofline (the path is compute-bound)
            gbps = traffic / kern_s / 1e9
            out["roofline"]["hbm"] = {"achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                      "frac": round(gbps / HBM_PEAK_GBPS, 4)}
        if strong is not None:
            out["strong"] = strong
        # the extra legs never decide whether the headline line gets printed
        if n_gpus == 1 and not fake and not a.no_d2h:
            try:
                out["d2h_inclusive"] = d2h_leg(g, d_code, d_f0, d_spkr, a.steps, audio_sec_per_step)
            except Exception as e:  # noqa: BLE001
                out["d2h_inclusive"] = {"error": f"{type(e).__name__}: {e}"}
        if n_gpus == 1 and not fake and not a.no_latency:
            try:
                out["latency"] = latency_leg(synth, g, dev)
            except Exception as e:  # noqa: BLE001
                out["latency"] = {"error": f"{type(e).__name__}: {e}"}
        if not a.no_split_bf16 and n_gpus == 1 and not fake:
            try:
                out["split_bf16"] = split_bf16_leg(synth, sd, dev, d_code, d_f0, d_spkr, y, a.steps,
                                                   audio_sec_per_step, flops_step)
            except Exception as e:  # noqa: BLE001
                out["split_bf16"] = {"error": f"{type(e).__name__}: {e}"}
        if not a.no_pipeline and n_gpus == 1 and not fake:
            try:
                out["pipeline"] = pipeline_leg(synth, dev, g)
            except Exception as e:  # noqa: BLE001
                out["pipeline"] = {"error": f"{type(e).__name__}: {e}"}
        y_split = out["split_bf16"].pop("_y_head", None) if isinstance(out.get("split_bf16"), dict) else None
        parity_failed = None
        if not a.no_cpu_baseline and n_gpus == 1:
            try:
                out["cpu_baseline"], ref_waves = cpu_baseline(synth, sd, torch.from_numpy(code), torch.from_numpy(f0),
                                                              torch.from_numpy(spkr))
                # parity in the same run: the timed GPU batch against the oracle's waveforms
                out["parity"] = parity_vs(ref_waves, y_head)
                if y_split is not None:
                    out["split_bf16"]["parity"] = parity_vs(ref_waves, y_split)
                pr = out["parity"]
                if not (pr["rms"] <= PARITY_TOL_RMS and pr["rel"] <= PARITY_TOL_REL and pr["guard"] <= 1.0):
                    parity_failed = (f"fp32 path: rms {pr['rms']:.3e} (tol {PARITY_TOL_RMS}, fp32 guard {PARITY_GUARD_RMS}), "
                                     f"rel {pr['rel']:.3e} (tol {PARITY_TOL_REL})")
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        elif not a.no_cpu_baseline and not fake:
            # N > 1: no CPU timing (rank 0, N = 1 only), but rank 0's timed batch is still checked against the oracle in the same run
            try:
                from oracle import generator_ref as gr
                wf = gr.fold_state_dict(sd)
                torch.set_num_threads(min(16, os.cpu_count() or 1))
                tc, tf, ts = torch.from_numpy(code), torch.from_numpy(f0), torch.from_numpy(spkr)
                ref_waves = [gr.code_generator(wf, synth.VCTK_CONFIG, tc[b:b + 1], tf[b:b + 1], ts[b:b + 1]) for b in range(2)]
                out["parity"] = parity_vs(ref_waves, y_head)
                pr = out["parity"]
                if not (pr["rms"] <= PARITY_TOL_RMS and pr["rel"] <= PARITY_TOL_REL and pr["guard"] <= 1.0):
                    parity_failed = f"fp32 path, rank 0 of {n_gpus}: rms {pr['rms']:.3e} (fp32 guard {PARITY_GUARD_RMS}), rel {pr['rel']:.3e}"
            except Exception as e:  # noqa: BLE001
                out["parity"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
        if dist is not None and not fake and out["rccl_ranks_seen"] != n_gpus:
            # --gpus N must mean N ranks in ONE process group: a launch that silently ran fewer (or separate) ranks is not a number
            print(f"bench.py: the process group reports {out['rccl_ranks_seen']} rank(s), --gpus {n_gpus} was asked for", file=sys.stderr,
                  flush=True)
            dist.destroy_process_group()
            sys.exit(4)
        if parity_failed:
            print("bench.py: PARITY FAILED -- " + parity_failed, file=sys.stderr, flush=True)
            if dist is not None:
                dist.destroy_process_group()
            sys.exit(3)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
