#!/usr/bin/env python
"""The strong-scaling job list of bench.py (1 024 ragged jobs of 2-5 s) through harness.run_resynthesis under different
batch caps:  python tools/strong_ab.py 64:24000 128:24000 128:32000  (max_batch:max_frames)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, dissc_amd, synthdata as synth
from dissc_amd import harness
from dissc_amd.generator import wav_postprocess_

dev = "cuda:0"
g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to(dev)
g.load_state_dict(synth.synth_generator_state_dict(seed=0))
g.eval().remove_weight_norm()
jobs = bench.strong_jobs(synth)
audio = sum(len(j["code"]) for j in jobs) * 320 / 16000.0
for spec in (sys.argv[1:] or ["64:24000"]):
    mb, mf = (int(v) for v in spec.split(":"))
    best = 1e9
    for rep in range(3):
        stats = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        harness.run_resynthesis(g, jobs, 0, 1, dev, None, max_batch=mb, max_frames=mf, postprocess=wav_postprocess_, stats=stats)
        best = min(best, time.perf_counter() - t0)
    print(f"max_batch {mb:4d} max_frames {mf:6d}: wall {best * 1e3:7.1f} ms = {audio / best:7.0f}x real time "
          f"(compute {stats['compute_s'] * 1e3:.1f} ms)", flush=True)
