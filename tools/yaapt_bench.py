#!/usr/bin/env python
"""YAAPT F0 of 32 x 10 s: device tracker vs the host-DP form.  python tools/yaapt_bench.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_yaapt import voiced
from dissc_amd.f0 import YaaptTracker
rs = np.random.RandomState(0)
sigs = []
for b in range(32):
    parts = []
    for k in range(10):
        parts.append(voiced(np.linspace(100 + 10 * k + b, 140 + 12 * k + b, 12000)))
        parts.append(0.003 * rs.standard_normal(4000))
    sigs.append(np.concatenate(parts).astype(np.float32))
trk = YaaptTracker("cuda:0")
for host in (False, True):
    trk(sigs, host_dp=host)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); out = trk(sigs, host_dp=host); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"host_dp={host}: {np.median(ts)*1e3:.1f} ms per 32 x 10 s = {320/np.median(ts):.0f}x real time", flush=True)
# stage times of the device path
from dissc_amd import f0 as f0m
pad = trk.flen // 2
lens = [len(x) + 2 * pad for x in sigs]
N = (max(lens) + 3) // 4 * 4
def T(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
def stage():
    wav = torch.zeros(len(sigs), N, dtype=torch.float32, pin_memory=True)
    for i, w in enumerate(sigs):
        wav[i, pad:pad + len(w)] = torch.as_tensor(np.asarray(w, dtype=np.float32))
    return wav
ms, wav = T(stage); print(f"host staging {ms:.1f} ms")
ms, s = T(lambda: trk.spectral(wav, torch.tensor(lens, dtype=torch.int32))); print(f"spectral {ms:.1f} ms")
nfr = [f0m.lib.dissc_yaapt_frames(trk._h, n) for n in lens]
ntd = [min(f0m.lib.dissc_yaapt_tda_frames(trk._h, n), f) for n, f in zip(lens, nfr)]
ms, st = T(lambda: trk.spec_track(s, nfr, ntd)); print(f"spec_track {ms:.1f} ms")
ms, c1 = T(lambda: trk.nccf(s["filt"], s["n_samples"], st["lag_min"], st["lag_max"])); print(f"nccf {ms:.1f} ms")
c2 = trk.nccf(s["nlfilt"], s["n_samples"], st["lag_min"], st["lag_max"])
ms, f0 = T(lambda: trk.final_track_device(st, c1, c2)); print(f"final_track {ms:.1f} ms")
ms, _ = T(lambda: f0.cpu().numpy()); print(f"d2h {ms:.1f} ms")
