#!/usr/bin/env python
"""Stage timings of the full pipeline (BASELINE configs[1]: 8 x 10 s utterances, encode ->
len/pitch prediction -> resynthesis, 1 GPU, in memory, synthetic weights)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dissc_amd  # noqa: E402
from dissc_amd import predictors as P  # noqa: E402
from dissc_amd.hubert import HubertEncoder  # noqa: E402
import synthdata as synth  # noqa: E402  (synthetic checkpoints / inputs only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = "cuda:0"
    n = int(a.seconds * 16000)
    enc = HubertEncoder(synth.synth_hubert_state_dict(6), synth.synth_kmeans_centers(), 6).to(dev)
    lm = P.LenPredictor(100, 108).to(dev)
    lm.load_state_dict(synth.synth_len_state_dict(100, 108))
    lm.norm_mean, lm.norm_std = synth.synth_len_norm_stats()
    pm = P.PitchPredictor(100, 108).to(dev)
    pm.load_state_dict(synth.synth_pitch_state_dict("new", 100, 108))
    g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to(dev)
    g.load_state_dict(synth.synth_generator_state_dict(0))
    g.eval().remove_weight_norm()
    wav = torch.stack([torch.from_numpy(synth.synth_waveform(n, seed=i)) for i in range(a.utts)]).to(dev)
    spk = [6] * a.utts  # p231

    def run():
        t = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = enc(wav, want_dense=False)
        torch.cuda.synchronize(); t["encode"] = time.perf_counter() - t0; t0 = time.perf_counter()
        units = out["units"]
        res = P.infer_samples([u for u in units.cpu()], spk, lm, pm, norm_pitch=True, device=dev)
        torch.cuda.synchronize(); t["infer"] = time.perf_counter() - t0; t0 = time.perf_counter()
        T = max(len(r[0]) for r in res)
        code = np.zeros((a.utts, T), np.int64); f0 = np.zeros((a.utts, 1, T), np.float32)
        lens = np.zeros(a.utts, np.int32)
        for i, (u, f, _) in enumerate(res):
            code[i, :len(u)] = u; f0[i, 0, :len(u)] = f; lens[i] = len(u)
        y = g(code=torch.from_numpy(code), f0=torch.from_numpy(f0), spkr=torch.tensor(spk).view(-1, 1),
              lengths=torch.from_numpy(lens))
        torch.cuda.synchronize(); t["resynth"] = time.perf_counter() - t0
        t["out_seconds"] = float(lens.sum()) * 320 / 16000
        return t

    run()
    ts = [run() for _ in range(a.iters)]
    med = {k: float(np.median([t[k] for t in ts])) for k in ts[0]}
    tot = med["encode"] + med["infer"] + med["resynth"]
    in_sec = a.utts * a.seconds
    # the same conversion through the device-resident Converter (dissc_amd/pipeline.py): wall time of the whole
    # call, host work included (batch assembly, one D2H of the packed waveforms), vs the sum of the stages above
    from dissc_amd.pipeline import Converter
    conv = Converter(enc, lm, pm, g)
    waves = [w for w in wav.cpu().numpy()]
    conv(waves, [6])
    wall = []
    for _ in range(a.iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = conv(waves, [6])
        wall.append(time.perf_counter() - t0)
    cw = float(np.median(wall))
    dwaves = [w for w in wav]  # the same utterances, already in HBM
    conv(dwaves, [6])
    wall_d = []
    for _ in range(a.iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        conv(dwaves, [6])
        wall_d.append(time.perf_counter() - t0)
    cwd = float(np.median(wall_d))
    print(json.dumps({"utts": a.utts, "seconds_each": a.seconds, "ms": {k: round(med[k] * 1e3, 2) for k in ("encode", "infer", "resynth")},
                      "total_ms": round(tot * 1e3, 2), "input_audio_sec_per_sec": round(in_sec / tot, 1),
                      "encode_x_realtime": round(in_sec / med["encode"], 1),
                      "resynth_x_realtime": round(med["out_seconds"] / med["resynth"], 1),
                      "converter_wall_ms": round(cw * 1e3, 2), "converter_host_overhead_frac": round(cw / tot - 1.0, 4),
                      "converter_wall_ms_inputs_in_hbm": round(cwd * 1e3, 2),
                      "converter_overhead_frac_inputs_in_hbm": round(cwd / tot - 1.0, 4),
                      "converter_outputs": len(out)}))


if __name__ == "__main__":
    main()
