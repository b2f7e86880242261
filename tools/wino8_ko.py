#!/usr/bin/env python
"""Knock-outs of conv_wino8_kernel (option wino8_dbg: bit 0 input transform, 1 MFMAs, 2 epilogue): where a layer's time
goes.  Results are wrong with any bit set; timing only."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dissc_amd._lib import lib, check  # noqa: E402
ms = ctypes.c_float()
shapes = ((256, 2500, 11, 1, 0, 4), (256, 2500, 11, 5, 0, 4), (128, 10000, 7, 1, 1, 4), (64, 40000, 11, 1, 1, 4), (128, 10000, 11, 3, 1, 4),
          (256, 2500, 7, 1, 0, 12), (256, 2500, 11, 1, 1, 12), (128, 10000, 7, 1, 1, 12), (64, 40000, 7, 1, 1, 4), (64, 40000, 7, 1, 1, 12),
          (64, 40000, 7, 3, 0, 4), (64, 40000, 7, 3, 0, 12), (64, 40000, 11, 5, 3, 4), (64, 40000, 11, 5, 3, 12))
for C, L, k, d, epi, flag in shapes:  # flag 4: F(6,3), 12: F(5,4)
    row = []
    for dbg in (0, 1, 2, 4, 3, 6, 7):
        assert lib.dissc_set_option(b"kernel_dbg", dbg) == 0
        check(lib.dissc_conv_bench(32, C, C, k, d, L, epi, 10, flag, ctypes.byref(ms)), "conv_bench")
        row.append(f"dbg={dbg}: {ms.value * 1e3:6.0f}")
    lib.dissc_set_option(b"kernel_dbg", 0)
    check(lib.dissc_conv_bench(32, C, C, k, d, L, epi, 10, 2, ctypes.byref(ms)), "conv_bench")
    print(f"C={C} L={L} k={k} d={d} epi={epi} {'F(5,4)' if flag == 12 else 'F(6,3)'} (us):  " + "  ".join(row) + f"   | F(4,3): {ms.value * 1e3:6.0f}", flush=True)
