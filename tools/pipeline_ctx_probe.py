#!/usr/bin/env python
"""Does what ran BEFORE bench.py's pipeline leg change its wall time?  The leg alone, then after each of the legs that precede it in a
full bench.py run (diagnostics: the full run read wall = stage sum + 2 ms, the leg alone wall = stage sum)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, dissc_amd
import synthdata as synth

dev = torch.device("cuda", 0)
sd = synth.synth_generator_state_dict(seed=0)
g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to(dev)
g.load_state_dict(sd); g.eval(); g.remove_weight_norm()
code, f0, spkr, _ = synth.synth_generator_inputs(32, 500, seed=0)
d_code, d_f0, d_spkr = (torch.from_numpy(v).to(dev) for v in (code, f0, spkr))
for _ in range(5):
    y = g(code=d_code, f0=d_f0, spkr=d_spkr)
torch.cuda.synchronize()


def show(tag):
    p = bench.pipeline_leg(synth, dev, g)
    print(f"{tag:28s} ms_per_batch {p['ms_per_batch']:.2f}  stage sum {p['stages']['sum_ms']:.2f}  stages {p['stages']['encode_ms']} / "
          f"{p['stages']['predict_ms']} / {p['stages']['generator_ms']} / {p['stages']['host_ms']}  reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB", flush=True)


show("alone")
bench.d2h_leg(g, d_code, d_f0, d_spkr, 10, 320.0)
bench.latency_leg(synth, g, dev)
bench.split_bf16_leg(synth, sd, dev, d_code, d_f0, d_spkr, y, 10, 320.0, g.flops(16000))
show("after split_bf16 leg")
torch.cuda.empty_cache()
show("after empty_cache")
bench.strong_leg(synth, g, dev, 0, 1, None)
show("after strong leg")
show("after strong leg, again")
torch.cuda.empty_cache()
show("after empty_cache")
