#!/usr/bin/env python
"""Tune the fused-ResBlock options on the GPU: ms/step of the bench workload per setting."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, json, torch
sys.path.insert(0, %r)
import dissc_amd
from dissc_amd._lib import lib
import synthdata as synth
opts = json.loads(sys.argv[1])
for k, v in opts.items():
    assert lib.dissc_set_option(k.encode(), int(v)) == 0
g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to("cuda:0")
g.load_state_dict(synth.synth_generator_state_dict(0)); g.eval().remove_weight_norm()
code, f0, spkr, _ = synth.synth_generator_inputs(32, 500, seed=1234)
c, f, s = torch.from_numpy(code).cuda(), torch.from_numpy(f0).cuda(), torch.from_numpy(spkr).cuda()
for _ in range(3): g(code=c, f0=f, spkr=s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): g(code=c, f0=f, spkr=s)
e1.record(); torch.cuda.synchronize()
print(json.dumps(opts), round(e0.elapsed_time(e1) / 10, 3), "ms/step")
''' % ROOT
import json
SETS = [json.loads(x) for x in sys.argv[1:]] or [{"wino": 1}, {"wino": 0}]
for opts in SETS:
    subprocess.run([sys.executable, "-c", CODE, json.dumps(opts)], check=False)
