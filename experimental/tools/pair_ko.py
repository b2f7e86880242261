#!/usr/bin/env python
"""Knock-outs of respair_wino_kernel (option wino_dbg: bit 0 input transforms, 1 MFMAs, 2 the A^T passes of the two
exchanges): where a tile's time goes.  Results are wrong with any bit set; timing only."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dissc_amd._lib import lib, check  # noqa: E402
ms = ctypes.c_float()
for C, L, k, d in ((32, 80000, 11, 1), (32, 80000, 7, 1), (64, 40000, 3, 1), (32, 80000, 11, 5)):
    row = []
    for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
        assert lib.dissc_set_option(b"kernel_dbg", dbg) == 0
        check(lib.dissc_pair_bench(32, C, k, d, L, 1, 10, 3, ctypes.byref(ms)), "pair_bench")
        row.append(f"dbg={dbg}: {ms.value * 1e3:6.0f}")
    lib.dissc_set_option(b"kernel_dbg", 0)
    print(f"C={C} k={k} d={d} (us):  " + "  ".join(row), flush=True)
