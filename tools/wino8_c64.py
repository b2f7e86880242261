#!/usr/bin/env python
"""The three C = 64 tile forms of conv_wino8_kernel (option wino8_c64_wide: 1 = 64 x 128 tiles, 0 = 64 x 64, 2 = 64 x 64 built for
two workgroups per CU) as F(6,3) and as F(5,4), per launch at the generator's shape, next to the F(4,3) kernel."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dissc_amd
from dissc_amd._lib import check
L = dissc_amd.lib
ms = ctypes.c_float()
for k in (7, 11):
    for d in (1, 3, 5):
        for epi in (0, 1, 3):
            row = []
            check(L.dissc_conv_bench(32, 64, 64, k, d, 40000, epi, 20, 2, ctypes.byref(ms)), "b"); row.append(f"F43 {ms.value*1e3:5.0f}")
            for name, flag in (("F63", 4), ("F54", 12)):
                for wide in (1, 0, 2):
                    check(L.dissc_set_option(b"wino8_c64_wide", wide), "o")
                    check(L.dissc_conv_bench(32, 64, 64, k, d, 40000, epi, 20, flag, ctypes.byref(ms)), "b")
                    row.append(f"{name} mode{wide} {ms.value*1e3:5.0f}")
            print(f"C64 k{k} d{d} epi{epi}: " + "  ".join(row), flush=True)
L.dissc_set_option(b"wino8_c64_wide", 3)
