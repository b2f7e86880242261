#!/usr/bin/env python
"""Generator cost per 16 000 frames as a function of the utterance length (uniform batches of 16 000 / T utterances):
what do short utterances (VCTK: 2-5 s = 100-250 frames) pay for per-utterance tiles?  Per-stage kernel time with
DISSC_OPTIONS=multistream=0 under rocprofv3 (tools/kstats.py) tells where."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dissc_amd
import synthdata as synth

g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to("cuda:0")
g.load_state_dict(synth.synth_generator_state_dict(seed=0))
g.eval().remove_weight_norm()
for T in [int(v) for v in (sys.argv[1:] or [500, 250, 200, 150, 125, 100, 64])]:
    B = 16000 // T
    code, f0, spkr, _ = synth.synth_generator_inputs(B, T, seed=5)
    c, f, s_ = (torch.from_numpy(x).cuda() for x in (code, f0, spkr))
    for _ in range(3):
        g(code=c, f0=f, spkr=s_)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g(code=c, f0=f, spkr=s_)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    print(f"T = {T:4d} x B = {B:4d}: {ms:6.2f} ms per forward, {ms * 16000 / (B * T):6.2f} ms per 16 000 frames", flush=True)
