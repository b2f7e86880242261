#!/usr/bin/env python
"""Generator latency / throughput vs batch size (T = 500 frames = 10 s)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dissc_amd  # noqa: E402
import synthdata as synth  # noqa: E402

g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to("cuda:0")
g.load_state_dict(synth.synth_generator_state_dict(0))
g.eval().remove_weight_norm()
print("| B | ms/forward | audio-sec/sec | TFLOP/s |")
print("|---|---|---|---|")
for B in [int(v) for v in os.environ.get("BATCHES", "1,2,4,8,16,32,64").split(",")]:
    code, f0, spkr, _ = synth.synth_generator_inputs(B, 500, seed=1)
    c, f, s = torch.from_numpy(code).cuda(), torch.from_numpy(f0).cuda(), torch.from_numpy(spkr).cuda()
    for _ in range(3):
        g(code=c, f0=f, spkr=s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        g(code=c, f0=f, spkr=s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"| {B} | {ms:.2f} | {B*10/ms*1e3:.0f} | {g.flops(B*500)/ms/1e9:.1f} |")
