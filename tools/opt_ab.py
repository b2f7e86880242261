#!/usr/bin/env python
"""A/B one generator option on the bench shape: ms per forward for each value, same process.
    python tools/opt_ab.py pair_max_c 0 16 32 [--batch 32 --frames 500 --iters 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dissc_amd  # noqa: E402
import synthdata as synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("key")
ap.add_argument("values", nargs="+", type=int)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--frames", type=int, default=500)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
def make_generator():
    g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to("cuda:0")
    g.load_state_dict(synth.synth_generator_state_dict(0))
    return g.eval().remove_weight_norm()


code, f0, spkr, _ = synth.synth_generator_inputs(a.batch, a.frames, seed=1234)
kw = dict(code=torch.from_numpy(code).cuda(), f0=torch.from_numpy(f0).cuda(), spkr=torch.from_numpy(spkr).cuda())
ref = None
for rep in range(2):
    for v in a.values:
        assert dissc_amd.lib.dissc_set_option(a.key.encode(), v) == 0
        g = make_generator()  # a handle snapshots the options when it is created (round 5): one generator per value
        for _ in range(3):
            y = g(**kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            y = g(**kw)
        e1.record()
        torch.cuda.synchronize()
        same = "" if ref is None else f" identical_to_first={bool(torch.equal(y, ref))}"
        if ref is None:
            ref = y.clone()
        print(f"{a.key}={v}: {e0.elapsed_time(e1) / a.iters:.3f} ms/forward{same}", flush=True)
