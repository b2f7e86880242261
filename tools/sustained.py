#!/usr/bin/env python
"""Sustained-clock check: ms/step of the bench workload in consecutive 20-step windows
(DISSC_OPTIONS selects the arithmetic mode).  usage: sustained.py [windows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dissc_amd  # noqa: E402
import synthdata as synth  # noqa: E402

g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to("cuda:0")
g.load_state_dict(synth.synth_generator_state_dict(0))
g.eval().remove_weight_norm()
code, f0, spkr, _ = synth.synth_generator_inputs(32, 500, seed=1234)
c, f, s = torch.from_numpy(code).cuda(), torch.from_numpy(f0).cuda(), torch.from_numpy(spkr).cuda()
for _ in range(3):
    g(code=c, f0=f, spkr=s)
torch.cuda.synchronize()
out = []
for w in range(int(sys.argv[1]) if len(sys.argv) > 1 else 15):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g(code=c, f0=f, spkr=s)
    e1.record()
    torch.cuda.synchronize()
    out.append(round(e0.elapsed_time(e1) / 20, 2))
print(os.environ.get("DISSC_OPTIONS", "default"), out)
