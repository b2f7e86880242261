#!/usr/bin/env python
"""What do ragged utterance lengths cost the generator?  32 utterances with 16 000 frames in total, lengths spread
uniformly in [500 - w, 500 + w]: ms per forward (the kernels skip tiles beyond an utterance's end, but tile and
workgroup-round quantisation and the shorter launch grids of the tail remain)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dissc_amd
import synthdata as synth

g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to("cuda:0")
g.load_state_dict(synth.synth_generator_state_dict(seed=0))
g.eval().remove_weight_norm()
rs = np.random.RandomState(0)
def lengths_for(spec):
    if spec[0] in "un":                           # u556: uniform
        return np.full(32, int(spec[1:]))
    mult = 4 if spec.startswith("m") else 1        # m100: ragged, every length a multiple of 4
    w = int(spec.lstrip("m"))
    lens = 500 + (rs.randint(-w, w + 1, size=32) if w else np.zeros(32, int))
    lens = lens // mult * mult
    lens[:16] = 1000 - lens[16:]                    # pairs that add up to 1 000 frames: 16 000 in total
    return lens


for spec in (sys.argv[1:] or ["0", "50", "100", "200"]):
    pad_to = int(spec[1:]) if spec.startswith("p") else 0   # p600: uniform 500 frames inside rows of 600
    lens = lengths_for("u500" if pad_to else spec)
    T = max(int(lens.max()), pad_to)
    code, f0, spkr, _ = synth.synth_generator_inputs(32, T, seed=5)
    c, f, s_ = (torch.from_numpy(x).cuda() for x in (code, f0, spkr))
    ln = None if spec.startswith("n") else torch.from_numpy(lens.astype(np.int32)).cuda()  # n500: no lengths tensor
    for _ in range(3):
        g(code=c, f0=f, spkr=s_, lengths=ln)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g(code=c, f0=f, spkr=s_, lengths=ln)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    print(f"spread {spec:>5s}: lengths {lens.min()}..{lens.max()} (sum {lens.sum()}), {ms:.2f} ms per forward, "
          f"{ms * 16000 / lens.sum():.2f} ms per 16 000 frames")
