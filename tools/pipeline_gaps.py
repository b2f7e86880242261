#!/usr/bin/env python
"""GPU idle gaps inside one Converter call: python tools/pipeline_gaps.py <kernel_trace.csv> (rocprofv3 --kernel-trace of
tools/pipeline_gaps.py --run)."""
import csv, sys, os
if len(sys.argv) > 1 and sys.argv[1] == "--run":
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
    pm = P.PitchPredictor(100, 108).to(dev); pm.load_state_dict(synth.synth_pitch_state_dict("new", 100, 108))
    g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to(dev); g.load_state_dict(synth.synth_generator_state_dict(0)); g.eval().remove_weight_norm()
    conv = Converter(enc, lm, pm, g)
    waves = [torch.from_numpy(synth.synth_waveform(160000, seed=i)).to(dev) for i in range(32)]
    for _ in range(3):
        conv(waves, [6])
    torch.cuda.synchronize()
    sys.exit(0)
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call = the kernels after the last large gap
st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
n = len(rows)
# find start of last call: largest 3 gaps split warm-up/calls; take the final segment beginning at a hubert_lengths kernel
starts = [i for i, r in enumerate(rows) if "hubert_lengths" in r["Kernel_Name"]]
i0 = starts[-1]
busy = sum(en[i] - st[i] for i in range(i0, n))
span = en[-1] - st[i0]
print(f"last call: {n - i0} kernels, span {span/1e6:.2f} ms, kernel busy {busy/1e6:.2f} ms, idle {(span-busy)/1e6:.2f} ms")
gaps = [(st[i + 1] - en[i], rows[i]["Kernel_Name"][:50], rows[i + 1]["Kernel_Name"][:50]) for i in range(i0, n - 1)]
tot_small = sum(g[0] for g in gaps if 0 < g[0] <= 20000)
print(f"sum of gaps <= 20 us: {tot_small/1e6:.2f} ms")
for g in sorted(gaps, reverse=True)[:15]:
    print(f"{g[0]/1e3:9.1f} us after {g[1]} -> {g[2]}")
