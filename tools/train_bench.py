#!/usr/bin/env python
"""Time one optimisation step of the three predictors at the reference's training batch
(batch_size 32, train_len_predictor.py:122 / train_f0_predictor.py:127) on device-resident inputs, with the CPU
restatement (oracle/train_ref.py, torch autograd on the host cores) timed beside it.
   python tools/train_bench.py [--L 200] [--steps 50] [--no-cpu]
prints one JSON line per model."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--L", type=int, default=200, help="padded units per utterance (deduped for the length model)")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    from dissc_amd.train import Trainer, init_state_dict
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    B, L = a.B, a.L
    n = rs.randint(L // 2, L + 1, size=B)
    n[0] = L
    seq = np.full((B, L), 100, np.int64)
    for b in range(B):
        seq[b, :n[b]] = rs.randint(0, 100, size=n[b])
    spk = rs.randint(0, 99, size=(B, 1)).astype(np.int64)
    stats = (torch.full((99,), 150.0), torch.full((99,), 30.0))
    for kind in ("len", "new", "base"):
        pad = -1.0 if kind == "len" else -100.0
        tgt = np.full((B, L), pad, np.float32)
        for b in range(B):
            tgt[b, :n[b]] = rs.randint(1, 9, size=n[b]) if kind == "len" else rs.randn(n[b]) * (rs.rand(n[b]) > 0.3)
        sd0 = init_state_dict(kind, 100, 99, seed=1)
        tr = Trainer(kind, sd0, 3e-4, norm=(3.0, 2.0), stats=stats).to(dev)
        keep, pm = tr.draw_masks(B, L)
        d = [torch.as_tensor(x).to(dev) for x in (seq, spk, tgt, keep)]
        pmd = None if pm is None else pm.to(dev)
        for _ in range(5):
            tr.step(d[0], d[1], d[2], keep=d[3], pe_mult=pmd)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            loss = tr.step(d[0], d[1], d[2], keep=d[3], pe_mult=pmd)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.steps
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = tr.step(d[0], d[1], d[2], keep=d[3], pe_mult=pmd)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3 / a.steps
        out = {"model": kind, "B": B, "L": L, "units": int(n.sum()), "gpu_ms_per_step": round(ms, 4),
               "wall_ms_per_step": round(wall, 4), "units_per_s": round(n.sum() / ms * 1e3, 1), "loss": float(loss)}
        if not a.no_cpu:
            from oracle import train_ref
            sd, st = {k: v.clone() for k, v in sd0.items()}, {}
            args = [torch.from_numpy(x) for x in (seq, spk, tgt)] + [keep]
            ts = []
            for _ in range(4):
                t0 = time.perf_counter()
                train_ref.train_step(kind, sd, *args, 3e-4, st, norm=(torch.tensor(3.0), torch.tensor(2.0)), stats=stats,
                                     pe_mult=pm)
                ts.append(time.perf_counter() - t0)
            out["cpu_ms_per_step"] = round(float(np.median(ts[1:])) * 1e3, 2)
            out["cpu_threads"] = torch.get_num_threads()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
