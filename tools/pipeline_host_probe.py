#!/usr/bin/env python
"""Where the wall time of one Converter call goes on the HOST side: perf_counter at every stage mark (dissc_amd/pipeline.py _mark) of
un-profiled calls, next to the HIP-event stage spans of profiled ones (bench.py's `pipeline` leg reports wall and the event spans)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dissc_amd, synthdata as synth
from dissc_amd import predictors as P
from dissc_amd.hubert import HubertEncoder
from dissc_amd.pipeline import Converter

dev = "cuda:0"
enc = HubertEncoder(synth.synth_hubert_state_dict(6), synth.synth_kmeans_centers(), 6).to(dev)
lm = P.LenPredictor(100, 108).to(dev); lm.load_state_dict(synth.synth_len_state_dict(100, 108))
lm.norm_mean, lm.norm_std = synth.synth_len_norm_stats()
pm = P.PitchPredictorBase(100, 108).to(dev); pm.load_state_dict(synth.synth_pitch_state_dict("base", 100, 108))
g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to(dev); g.load_state_dict(synth.synth_generator_state_dict(0)); g.eval().remove_weight_norm()
conv = Converter(enc, lm, pm, g)
waves = [torch.from_numpy(synth.synth_waveform(160000, seed=i)).to(dev) for i in range(32)]
for _ in range(3):
    conv(waves, [6])
torch.cuda.synchronize()
stamps = []
orig = Converter._mark
def mark(self, marks, name):
    stamps.append((name, time.perf_counter()))
    return orig(self, marks, name)
Converter._mark = mark
rows = []
for _ in range(7):
    stamps.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = conv(waves, [6])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append([("call", 0.0)] + [(n, (t - t0) * 1e3) for n, t in stamps] + [("return", (t1 - t0) * 1e3), ("synced", (t2 - t0) * 1e3)])
med = rows[len(rows) // 2]
print("host timeline of one call (ms since the call, median-ish run): " + "  ".join(f"{n} {t:.2f}" for n, t in med))
print("wall per call (ms): " + " ".join(f"{r[-1][1]:.2f}" for r in rows))
