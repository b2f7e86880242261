#!/usr/bin/env python
"""Rehearse the strong-scaling list (bench.py `strong`: 1 024 ragged jobs, the same list at every N) at N = 1, 2, 4, 8
ranks on ONE MI355X and turn the measurements into a prediction for the N-GPU node the builder cannot reach
(VERDICT r03, next-round item 1).

    python tools/strong_rehearsal.py [--out profiles/r04] [--ns 1 2 4 8]

Per N it launches `bench.py --gpus N --strong-only` under torch.distributed.run with DISSC_BENCH_BACKEND=gloo (the ranks
share the GPU; gloo stages the exchange through the host) and DISSC_STRONG_SOLO=1, and takes from each run:

  * `solo`: every rank's share -- the SAME rounds and batches as in the N-rank run -- computed with the device to
    itself (host batching + H2D + generator + post-processing + pack; the others wait at a barrier).  This carries the
    real costs of sharding: LPT imbalance, smaller batches per round, per-round fixed costs.
  * at N = 1 (a real run, nothing shared): the wall T_1, and rank 0's delivery cost for the WHOLE list
    D = device-to-host copy + unpack + sink (rank 0 receives every waveform at any N, so D does not depend on N).
  * the contended gloo wall (functional rehearsal of the N-rank path; recorded, not used by the model).

Model: rounds of one run are pipelined -- round k is delivered while round k + 1 computes -- so

    T_N = max_r C_r  +  G_last + D / R_N            (exposed tail = the last round's gather + delivery)
    T_N >= C_first + sum G + D                       (delivery-bound floor: rank 0 cannot deliver faster than D)

with C_r the solo wall of rank r, R_N the rounds of the run, G a modelled RCCL all-gather of the round's exchange
buffers at XGMI_RING_GBPS (bench.py; conservative single-ring figure).  Efficiency_N = T_1 / (N * T_N).
Writes <out>/strong_model.json (read back by bench.py -> `strong.predicted`) and <out>/strong_model.md.
"""
import argparse
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_n(n, timeout):
    env = dict(os.environ, DISSC_BENCH_BACKEND="gloo", DISSC_STRONG_SOLO="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if n == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--strong-only"]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
               "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n),
               "--strong-only"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    for line in reversed(r.stdout.splitlines()):
        if line.startswith("{") and '"strong"' in line:
            return json.loads(line)
    raise RuntimeError(f"N={n}: no strong line (rc {r.returncode})\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}")


def model(runs, xgmi_gbps):
    one = runs[1]["strong"]
    t1 = one["wall_ms"]
    d_total = one["rank0_delivery"]["d2h_unpack_ms"] + one["rank0_delivery"]["sink_ms"]
    payload = one["exchange"]["payload_bytes_this_rank"]  # N = 1: the whole list
    per_n = {}
    for n, rec in sorted(runs.items()):
        st = rec["strong"]
        solo = st["solo"]["wall_ms"]
        rounds = st["exchange"]["rounds"]
        sent = st["exchange"]["sent_bytes_per_rank"]          # summed over the rounds, per rank
        g_total = sent * (n - 1) / (xgmi_gbps * 1e9) * 1e3 if n > 1 else 0.0   # ring: every rank receives (N-1) buffers
        g_last = g_total / rounds
        c_max = max(solo)
        tail = g_last + d_total / rounds
        t_pipe = c_max + tail
        t_floor = c_max / rounds + g_total + d_total           # delivery-bound floor
        t_n = max(t_pipe, t_floor)
        per_n[str(n)] = {
            "predicted_wall_ms": round(t_n, 2), "predicted_speedup": round(t1 / t_n, 3),
            "predicted_efficiency": round(t1 / t_n / n, 3), "rounds": rounds,
            "solo_compute_ms": solo, "max_solo_compute_ms": round(c_max, 2),
            "mean_solo_compute_ms": round(sum(solo) / len(solo), 2),
            "exposed_tail_ms": round(tail, 2), "modelled_allgather_ms_total": round(g_total, 2),
            "delivery_ms_total": round(d_total, 2), "delivery_bound": bool(t_floor > t_pipe),
            "rehearsal_gloo_wall_ms": st["wall_ms"], "load_imbalance": st["load_imbalance"],
        }
    return {"what": "strong-scaling prediction from one-GPU rehearsals (tools/strong_rehearsal.py; model: DESIGN.md section 7)",
            "jobs": one["jobs"], "audio_sec": one["audio_sec"], "payload_bytes": payload,
            "t1_measured_wall_ms": t1, "xgmi_allgather_gbps_assumed": xgmi_gbps, "per_n": per_n}


def markdown(m):
    rows = ["| N | rounds | max solo compute (ms) | mean solo (ms) | LPT imbalance | modelled all-gather (ms) | exposed tail (ms) | "
            "predicted wall (ms) | speed-up | efficiency | rehearsal wall, gloo on one GPU (ms) |",
            "|---|---|---|---|---|---|---|---|---|---|---|"]
    for n, r in sorted(m["per_n"].items(), key=lambda kv: int(kv[0])):
        rows.append(f"| {n} | {r['rounds']} | {r['max_solo_compute_ms']} | {r['mean_solo_compute_ms']} | {r['load_imbalance']} | "
                    f"{r['modelled_allgather_ms_total']} | {r['exposed_tail_ms']} | {r['predicted_wall_ms']} | "
                    f"{r['predicted_speedup']} | {r['predicted_efficiency']} | {r['rehearsal_gloo_wall_ms']} |")
    head = (f"Strong-scaling prediction: {m['jobs']} jobs, {m['audio_sec']} s of audio, T_1 measured {m['t1_measured_wall_ms']} ms; "
            f"rank 0's delivery of the whole list {next(iter(m['per_n'].values()))['delivery_ms_total']} ms; all-gather modelled at "
            f"{m['xgmi_allgather_gbps_assumed']} GB/s per rank.\n\n")
    return head + "\n".join(rows) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04"))
    ap.add_argument("--ns", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--timeout", type=int, default=900)
    a = ap.parse_args()
    import bench
    runs = {}
    for n in a.ns:
        runs[n] = run_n(n, a.timeout)
        st = runs[n]["strong"]
        print(f"N={n}: wall {st['wall_ms']} ms (gloo, shared GPU), solo {st['solo']['wall_ms']}, rounds {st['exchange']['rounds']}",
              flush=True)
    if 1 not in runs:
        raise SystemExit("the model needs the N = 1 run")
    m = model(runs, bench.XGMI_RING_GBPS)
    m["runs"] = {str(n): r for n, r in runs.items()}
    os.makedirs(a.out, exist_ok=True)
    with open(os.path.join(a.out, "strong_model.json"), "w") as f:
        json.dump(m, f, indent=1)
    md = markdown(m)
    with open(os.path.join(a.out, "strong_model.md"), "w") as f:
        f.write(md)
    print(md)


if __name__ == "__main__":
    main()
