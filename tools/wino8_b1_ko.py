#!/usr/bin/env python
"""conv_wino8_kernel on the small-grid tiers (B = 1): per-launch time and its knock-outs (option kernel_dbg: bit 0 input
transform, 1 MFMAs, 2 epilogue; latency form also 3 window staging, 4 the round barrier, 5 weight loads).  Results are wrong
with any bit set; timing only."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dissc_amd._lib import lib, check  # noqa: E402
ms = ctypes.c_float()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
shapes = ((256, 2500, 11, 1, 0, 12), (256, 2500, 7, 1, 0, 12), (128, 10000, 11, 1, 1, 12), (64, 40000, 11, 1, 1, 4), (64, 40000, 7, 1, 1, 12))
for C, L, k, d, epi, flag in shapes:  # flag 4: F(6,3), 12: F(5,4)
    row = []
    for dbg in (0, 4, 4 + 1, 4 + 2, 4 + 8, 4 + 16, 4 + 32, 4 + 3, 4 + 3 + 8, 4 + 3 + 8 + 16, 63):
        assert lib.dissc_set_option(b"kernel_dbg", dbg) == 0
        check(lib.dissc_conv_bench(B, C, C, k, d, L, epi, 200, flag, ctypes.byref(ms)), "conv_bench")
        row.append(f"{dbg}: {ms.value * 1e3:5.1f}")
    lib.dissc_set_option(b"kernel_dbg", 0)
    print(f"B={B} C={C} L={L} k={k} d={d} epi={epi} {'F(5,4)' if flag == 12 else 'F(6,3)'} (us; dbg bits):  " + "  ".join(row), flush=True)
