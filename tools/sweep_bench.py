#!/usr/bin/env python
"""BASELINE configs[3]/[4]-style workload on ONE GPU: a many-to-many sweep of ragged utterances
(ESD-like, 2-5 s) x 4 target speakers through the resynthesis harness (LPT shard -> length-bucketed
batches -> generator -> GPU post-processing -> packed gather).  A sample of the jobs is re-run one
utterance at a time and must reproduce the batched waveforms bit for bit (the parity of the B=1
path against the reference is what tests/ checks).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dissc_amd  # noqa: E402
import synthdata as synth  # noqa: E402
from dissc_amd import harness  # noqa: E402


def main():
    n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    targets = [6, 0, 57, 101]
    rs = np.random.RandomState(3)
    jobs = []
    for u in range(n_utts):
        T = int(rs.randint(100, 251))
        code, f0, _, _ = synth.synth_generator_inputs(1, T, seed=5000 + u)
        for t in targets:
            jobs.append(dict(code=code[0], f0=f0[0, 0], spkr=t))
    sd = synth.synth_generator_state_dict(0)
    g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to("cuda:0")
    g.load_state_dict(sd)
    g.eval().remove_weight_norm()
    harness.run_resynthesis(g, jobs, device="cuda:0")  # warm-up (steady state of a long sweep: staging buffers exist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    waves = harness.run_resynthesis(g, jobs, device="cuda:0")
    dt = time.perf_counter() - t0
    audio = sum(len(j["code"]) for j in jobs) * 320 / 16000.0
    mismatches = 0
    for k in rs.choice(len(jobs), 6, replace=False):
        one = harness.run_resynthesis(g, [jobs[int(k)]], device="cuda:0")[0]
        mismatches += int(not np.array_equal(one, waves[int(k)]))
    print(json.dumps({"jobs": len(jobs), "utterances": n_utts, "targets": len(targets), "audio_sec": round(audio, 1),
                      "wall_s": round(dt, 3), "audio_sec_per_sec_incl_host": round(audio / dt, 1),
                      "batched_vs_single_mismatches_on_6_jobs": mismatches}))


if __name__ == "__main__":
    main()
