#!/usr/bin/env python
"""Knock-out timings of conv_wino_kernel (option wino_dbg: 1 no transform, 2 no MFMAs, 4 no epilogue, 8 no staging)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dissc_amd
from dissc_amd._lib import check
L = dissc_amd.lib
shapes = [(256, 11, 1, 2500), (256, 7, 1, 2500), (256, 3, 1, 2500), (128, 11, 5, 10000)]
if os.environ.get('WINO_SHAPES'):
    shapes = [tuple(int(v) for v in t.split('x')) for t in os.environ['WINO_SHAPES'].split(',')]
for C, k, d, Ln in shapes:
  for cpr in (32,):
    check(L.dissc_set_option(b"wino_cpr", cpr), "opt")
    out = [f"cpr{cpr}"]
    for dbg in [int(x) for x in os.environ.get('WINO_DBGS', '0,1,2,4,8,13,6,15').split(',')]:
        check(L.dissc_set_option(b"kernel_dbg", dbg), "opt")
        ms = ctypes.c_float()
        best = 1e9
        for rep in range(2):
            check(L.dissc_conv_bench(int(os.environ.get("WINO_B", "32")), C, C, k, d, Ln, int(os.environ.get("WINO_EPI", "0")), 20, 2, ctypes.byref(ms)), "bench")
            best = min(best, ms.value)
        out.append(f"dbg{dbg}: {best*1e3:6.0f}")
    print(f"C{C} k{k} d{d} (us): " + " | ".join(out), flush=True)
