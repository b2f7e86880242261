#!/usr/bin/env python
"""small-batch generator forwards with and without hipGraph replay (DISSC_EXPERIMENTAL=1 builds only): python tools/graph_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dissc_amd, synthdata as synth
def make(graphs):  # a handle snapshots the options when it is created (round 5): one generator per setting
    assert dissc_amd.lib.dissc_set_option(b"graphs", graphs) == 0
    g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to("cuda:0"); g.load_state_dict(synth.synth_generator_state_dict(0)); g.eval().remove_weight_norm()
    c, f, s, _ = synth.synth_generator_inputs(1, 3, seed=1)
    g(code=torch.from_numpy(c), f0=torch.from_numpy(f), spkr=torch.from_numpy(s))  # builds the native handle under the option
    return g
gs = {0: make(0), 1: make(1)}
dissc_amd.lib.dissc_set_option(b"graphs", 0)
for B, T in [(1, 100), (1, 500), (2, 500), (4, 500), (8, 250)]:
    code, f0, spkr, _ = synth.synth_generator_inputs(B, T, seed=1234)
    kw = dict(code=torch.from_numpy(code).cuda(), f0=torch.from_numpy(f0).cuda(), spkr=torch.from_numpy(spkr).cuda())
    if B > 1:
        kw["lengths"] = torch.tensor([T - 7 * i for i in range(B)], dtype=torch.int32).cuda()
    res = {}
    for v in (0, 1, 0, 1):
        g = gs[v]
        for _ in range(5): y = g(**kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): y = g(**kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        res.setdefault(v, []).append(ms)
        if v == 0: ref = y.clone()
        else: assert torch.equal(y, ref), "graph replay differs"
    import ctypes
    h, c = ctypes.c_int(), ctypes.c_int()
    dissc_amd.lib.dissc_get_option(b"graph_hits", ctypes.byref(h)); dissc_amd.lib.dissc_get_option(b"graph_captures", ctypes.byref(c))
    print(f"   hits {h.value} captures {c.value}")
    print(f"B={B} T={T}: plain {min(res[0]):.3f} ms, graph {min(res[1]):.3f} ms  ({B*T*0.02/min(res[1])*1e3:.0f}x real time)", flush=True)
