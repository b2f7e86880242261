#!/usr/bin/env python
"""Would two half-batch forwards running concurrently (two handles, two streams) beat one full-batch forward?
The stages differ in what bounds them (s0 / s1: MFMA, s2 k = 3 / s3 / s4: latency); two pipelines half a stage apart
might fill each other's gaps.  B = 32 x 10 s as 1 x 32, 2 x 16 concurrent, 4 x 8 concurrent."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dissc_amd
import synthdata as synth

dev = "cuda:0"
sd = synth.synth_generator_state_dict(seed=0)
code, f0, spkr, _ = synth.synth_generator_inputs(32, 500, seed=1234)
code, f0, spkr = (torch.from_numpy(v).to(dev) for v in (code, f0, spkr))


def make(n):
    gs = []
    for _ in range(n):
        g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to(dev)
        g.load_state_dict(sd)
        g.eval().remove_weight_norm()
        gs.append(g)
    return gs


for parts in (1, 2, 4):
    gs = make(parts)
    streams = [torch.cuda.Stream() for _ in range(parts)]
    bs = 32 // parts

    def step():
        outs = []
        cur = torch.cuda.current_stream()
        for i, (g, s) in enumerate(zip(gs, streams)):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(g(code=code[i * bs:(i + 1) * bs], f0=f0[i * bs:(i + 1) * bs], spkr=spkr[i * bs:(i + 1) * bs]))
        for s in streams:
            cur.wait_stream(s)
        return outs
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    print(f"{parts} x {bs}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per 32 x 10 s", flush=True)
    del gs
