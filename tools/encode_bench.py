#!/usr/bin/env python
"""HuBERT unit-encode timing: B x seconds of synthetic audio through dissc_amd.hubert.HubertEncoder.
    python tools/encode_bench.py [--utts 32 --seconds 10 --iters 10]
Prints one JSON line (ms per batch, x real time, algorithmic TFLOP/s)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dissc_amd.hubert import HubertEncoder  # noqa: E402
import synthdata as synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--utts", type=int, default=32)
ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
n = int(a.seconds * 16000)
enc = HubertEncoder(synth.synth_hubert_state_dict(6), synth.synth_kmeans_centers(), 6).to("cuda:0")
wav = torch.stack([torch.from_numpy(synth.synth_waveform(n, seed=i)) for i in range(a.utts)]).cuda()
for _ in range(3):
    out = enc(wav, want_dense=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    out = enc(wav, want_dense=False)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
# algorithmic FLOPs per utterance (conv feature extractor 49.1 + encoder 52.1 GFLOP per 10 s)
gflop = 101.2 * a.seconds / 10.0 * a.utts
print(json.dumps({"utts": a.utts, "seconds": a.seconds, "ms_per_batch": round(ms, 3),
                  "x_realtime": round(a.utts * a.seconds / ms * 1e3, 1), "tflops": round(gflop / ms, 1),
                  "units_checksum": int(out["units"].sum())}))
