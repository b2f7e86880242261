#!/usr/bin/env python
"""Knock-outs of respair32_f23_kernel (option wino_dbg: bit 0 the tap loops, 1 the T epilogue, 2 the output epilogue): where a
pair's time goes.  Results are wrong with any bit set; timing only.  Run with DISSC_OPTIONS=pair_f23=1."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dissc_amd._lib import lib, check  # noqa: E402
ms = ctypes.c_float()
C = int(os.environ.get("F23_C", "32"))
K = int(os.environ.get("F23_K", "11"))
for d, epi in ((1, 1), (5, 3)):
    row = []
    for dbg in (0, 1, 2, 4, 3, 7):
        assert lib.dissc_set_option(b"kernel_dbg", dbg) == 0
        check(lib.dissc_pair_bench(32, C, K, d, 80000 * 32 // C, epi, 20, int(os.environ.get("F23_MODE", "3")), ctypes.byref(ms)), "pair_bench")
        row.append(f"dbg={dbg}: {ms.value * 1e3:6.0f}")
    lib.dissc_set_option(b"kernel_dbg", 0)
    print(f"C={C} k={K} d={d} epi={epi} (us):  " + "  ".join(row), flush=True)
