#!/usr/bin/env python
"""A/B one option on single generator layers (dissc_conv_bench): python tools/conv_ab.py wdepth 1 2"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dissc_amd
L = dissc_amd.lib
key, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
shapes = [(256, 3, 1, 2500), (256, 7, 1, 2500), (256, 11, 1, 2500), (128, 3, 1, 10000), (128, 11, 5, 10000), (64, 3, 1, 40000), (64, 7, 3, 40000), (64, 11, 1, 40000)]
for C, k, d, Ln in shapes:
    out = []
    for rep in range(2):
        for v in vals:
            assert L.dissc_set_option(key.encode(), v) == 0
            ms = ctypes.c_float()
            rc = L.dissc_conv_bench(32, C, C, k, d, Ln, 0, 20, 0, ctypes.byref(ms))
            assert rc == 0, L.dissc_last_error()
            out.append(f"{key}={v}: {ms.value*1e3:7.1f} us {2*C*C*k*Ln*32/ms.value/1e9:6.1f} TF")
    print(f"C{C} k{k} d{d}: " + " | ".join(out), flush=True)
