#!/usr/bin/env python
"""Resynthesis rate including the GPU post-processing and the D2H copy of the waveforms
(the figure DESIGN.md quotes next to bench.py's HBM-resident `value`)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dissc_amd  # noqa: E402
import synthdata as synth  # noqa: E402
from dissc_amd.generator import wav_postprocess_  # noqa: E402

g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to("cuda:0")
g.load_state_dict(synth.synth_generator_state_dict(0))
g.eval().remove_weight_norm()
code, f0, spkr, _ = synth.synth_generator_inputs(32, 500, seed=1234)
n = torch.full((32,), 160000, dtype=torch.int32)
host = torch.empty(32, 1, 160000, pin_memory=True)
for it in range(3):
    y = g(code=torch.from_numpy(code), f0=torch.from_numpy(f0), spkr=torch.from_numpy(spkr))
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for it in range(K):
    y = g(code=torch.from_numpy(code), f0=torch.from_numpy(f0), spkr=torch.from_numpy(spkr))  # H2D of the codes included
    wav_postprocess_(y, n)
    host.copy_(y, non_blocking=True)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"H2D codes + generator + postprocess + D2H (pinned): {dt*1e3:.2f} ms/step -> {320/dt:.0f} audio-sec/sec")
