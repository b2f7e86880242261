#!/usr/bin/env python
"""B = 1 generator forward: wall per forward vs sum of kernel durations (rocprofv3 --kernel-trace of `--run`)."""
import csv, sys, os
if len(sys.argv) > 1 and sys.argv[1] == "--run":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import dissc_amd, synthdata as synth
    g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to("cuda:0"); g.load_state_dict(synth.synth_generator_state_dict(0)); g.eval().remove_weight_norm()
    code, f0, spkr, _ = synth.synth_generator_inputs(1, 500, seed=1234)
    kw = dict(code=torch.from_numpy(code).cuda(), f0=torch.from_numpy(f0).cuda(), spkr=torch.from_numpy(spkr).cuda())
    for _ in range(5): y = g(**kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = g(**kw)
    e1.record(); torch.cuda.synchronize()
    print("B=1 T=500: %.3f ms per forward" % (e0.elapsed_time(e1) / 20))
    sys.exit(0)
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
emb = [i for i, r in enumerate(rows) if "embed_concat" in r["Kernel_Name"]]
a, b = emb[-2], emb[-1]
seg = rows[a:b]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
# union of busy intervals (streams overlap)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
u = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: u += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
u += ce - cs
print(f"{len(seg)} kernels per forward, period {span/1e3:.0f} us, kernel-time sum {busy/1e3:.0f} us, GPU non-idle (union) {u/1e3:.0f} us, idle {(span-u)/1e3:.0f} us")
