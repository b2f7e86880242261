#!/usr/bin/env python
"""Rehearse the strong-scaling list (bench.py `strong`: 1 024 ragged jobs, the same list at every N) for N = 1, 2, 4, 8
ranks on ONE MI355X and turn the measurements into a prediction for the N-GPU node the builder cannot reach
(VERDICT r03, next-round item 1).

    python tools/strong_rehearsal.py [--out profiles/r04] [--ns 1 2 4 8] [--gloo-ranks 2]

One process runs `bench.py --strong-only` with DISSC_STRONG_EMULATE=<ns> and yields:

  * the real N = 1 run: the wall T_1 and rank 0's delivery cost for the WHOLE list, D = device-to-host copy + unpack +
    sink (rank 0 receives every waveform at any N, so D does not depend on N);
  * `emulated[N]`: every rank's share of an N-rank run -- the same LPT partition, rounds and batches -- computed alone,
    one share after the other (host batching + H2D + generator + post-processing + pack).  This carries the real
    costs of sharding: LPT imbalance, smaller batches per round, per-round fixed costs.
    (N processes sharing the one GPU do NOT give these numbers: with 4 or 8 processes the hardware queues are
    oversubscribed and every share takes 180-290 ms whatever its size; the first version of this tool measured that.)
  * optionally (--gloo-ranks R) the R-rank path for real, R gloo ranks sharing the GPU: a functional rehearsal whose
    contended wall is recorded, not used.

Model: rounds of one run are pipelined -- round k is delivered while round k + 1 computes -- so

    T_N = max_r C_r  +  f_last (G + D)              (exposed tail = the last round's gather + delivery; f_k = round k's
                                                      share of the payload -- rounds taper, the last is the smallest)
    T_N >= f_first max_r C_r + G + D                 (delivery-bound floor: rank 0 cannot deliver faster than D)

with C_r the solo wall of rank r's share, R_N the rounds of the run, G a modelled RCCL all-gather of the round's
exchange buffers at XGMI_RING_GBPS (bench.py; conservative single-ring figure).  Efficiency_N = T_1 / (N * T_N), T_1 the
model's own one-rank wall (the measured one is printed next to it).
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


def run_bench(n, timeout, env_extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
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


def model(one, xgmi_gbps):
    t1 = one["wall_ms"]
    t1_model = t1
    d_total = one["rank0_delivery"]["d2h_unpack_ms"] + one["rank0_delivery"]["sink_ms"]
    per_n = {}
    for key, e in sorted(((k, v) for k, v in one["emulated"].items() if k != "what"), key=lambda kv: int(kv[0])):
        n = int(key)
        solo, rounds, sent = e["share_wall_ms"], e["rounds"], e["sent_bytes_per_rank"]
        g_total = sent * (n - 1) / (xgmi_gbps * 1e9) * 1e3 if n > 1 else 0.0   # ring: every rank receives (N-1) buffers
        frac = e.get("round_fractions") or [1.0 / rounds] * rounds   # rounds taper: the last one is the smallest
        c_max = max(solo)
        tail = (g_total + d_total) * frac[-1]
        t_pipe = c_max + tail
        t_floor = c_max * frac[0] + g_total + d_total          # delivery-bound floor
        t_n = max(t_pipe, t_floor)
        if n == 1:
            t1_model = t_n   # speed-ups are taken against the MODEL's one-rank wall: the measured one carries that run's host noise
        per_n[key] = {
            "predicted_wall_ms": round(t_n, 2), "predicted_speedup": round(t1_model / t_n, 3),
            "predicted_efficiency": round(t1_model / t_n / n, 3), "rounds": rounds,
            "solo_compute_ms": solo, "max_solo_compute_ms": round(c_max, 2),
            "mean_solo_compute_ms": round(sum(solo) / len(solo), 2),
            "sum_solo_compute_ms": round(sum(solo), 2),
            "exposed_tail_ms": round(tail, 2), "modelled_allgather_ms_total": round(g_total, 2),
            "delivery_ms_total": round(d_total, 2), "delivery_bound": bool(t_floor > t_pipe),
            "load_imbalance": e["load_imbalance"], "round_fractions": frac,
        }
    return {"what": "strong-scaling prediction from a one-GPU rehearsal (tools/strong_rehearsal.py; model: DESIGN.md section 7)",
            "jobs": one["jobs"], "audio_sec": one["audio_sec"], "payload_bytes": one["exchange"]["payload_bytes_this_rank"],
            "t1_measured_wall_ms": t1, "t1_exposed_tail_ms": one["exposed_tail_ms"],
            "xgmi_allgather_gbps_assumed": xgmi_gbps, "per_n": per_n}


def markdown(m):
    rows = ["| N | rounds | max share alone (ms) | mean share (ms) | sum of shares (ms) | LPT imbalance | modelled all-gather (ms) | "
            "exposed tail (ms) | predicted wall (ms) | speed-up | efficiency |",
            "|---|---|---|---|---|---|---|---|---|---|---|"]
    for n, r in sorted(m["per_n"].items(), key=lambda kv: int(kv[0])):
        rows.append(f"| {n} | {r['rounds']} | {r['max_solo_compute_ms']} | {r['mean_solo_compute_ms']} | {r['sum_solo_compute_ms']} | "
                    f"{r['load_imbalance']} | {r['modelled_allgather_ms_total']} | {r['exposed_tail_ms']} | "
                    f"{r['predicted_wall_ms']} | {r['predicted_speedup']} | {r['predicted_efficiency']} |")
    head = (f"Strong-scaling prediction: {m['jobs']} jobs, {m['audio_sec']} s of audio, T_1 measured {m['t1_measured_wall_ms']} ms "
            f"(exposed tail {m['t1_exposed_tail_ms']} ms); rank 0's delivery of the whole list "
            f"{next(iter(m['per_n'].values()))['delivery_ms_total']} ms; all-gather modelled at "
            f"{m['xgmi_allgather_gbps_assumed']} GB/s per rank.\n\n")
    tail = ""
    if m.get("gloo_rehearsal"):
        g = m["gloo_rehearsal"]
        tail = (f"\nFunctional rehearsal, {g['ranks']} gloo ranks sharing the one GPU: wall {g['wall_ms']} ms (rank 0 delivers), "
                f"{g['own_rows_wall_ms']} ms (every rank delivers its rows); contended, not used by the model.\n")
    return head + "\n".join(rows) + "\n" + tail


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04"))
    ap.add_argument("--ns", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--gloo-ranks", type=int, default=2, help="also run this many gloo ranks on the one GPU (0 = skip)")
    ap.add_argument("--timeout", type=int, default=1200)
    a = ap.parse_args()
    import bench
    rec = run_bench(1, a.timeout, {"DISSC_STRONG_EMULATE": ",".join(str(n) for n in a.ns)})
    one = rec["strong"]
    m = model(one, bench.XGMI_RING_GBPS)
    if a.gloo_ranks > 1:
        g = run_bench(a.gloo_ranks, a.timeout, {"DISSC_BENCH_BACKEND": "gloo"})["strong"]
        m["gloo_rehearsal"] = {"ranks": a.gloo_ranks, "wall_ms": g["wall_ms"], "own_rows_wall_ms": g["own_rows"]["wall_ms"],
                               "per_rank_compute_ms": g["per_rank_compute_ms"], "exchange": g["exchange"]}
    m["n1_run"] = one
    os.makedirs(a.out, exist_ok=True)
    with open(os.path.join(a.out, "strong_model.json"), "w") as f:
        json.dump(m, f, indent=1)
    md = markdown(m)
    with open(os.path.join(a.out, "strong_model.md"), "w") as f:
        f.write(md)
    print(md)


if __name__ == "__main__":
    main()
