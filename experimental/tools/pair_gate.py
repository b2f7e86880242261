#!/usr/bin/env python
"""Gate experiment for respair_wino.hip: one residual pair per launch with both convs in the
Toom-Cook transform domain, against what runs today -- the direct fused pair (C = 32) or two conv_wino launches
(C = 64) -- through dissc_pair_bench at the generator's shapes (B = 32 x 10 s).
    python tools/pair_gate.py [iters]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dissc_amd._lib import lib, check  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ms = ctypes.c_float()
NAMES = {0: "two direct", 1: "fused direct", 2: "two wino", 3: "FUSED WINO"}
for C, L, shapes in ((32, 80000, [(7, 1), (7, 3), (7, 5), (11, 1), (11, 3), (11, 5)]), (64, 40000, [(3, 1), (3, 3), (3, 5)])):
    for k, d in shapes:
        for epi in (1, 3):
            row = []
            for mode, chv in (((1, 1), (3, 2), (3, 1)) if C == 32 else ((2, 1), (3, 2), (3, 1))):
                assert lib.dissc_set_option(b"pairw_chv", chv) == 0
                check(lib.dissc_pair_bench(32, C, k, d, L, epi, iters, mode, ctypes.byref(ms)), f"pair_bench mode {mode}")
                ns = (k + 2) // 3
                gf_alg = 2 * 2.0 * C * C * k * L * 32 / 1e9
                gf_exec = gf_alg if mode in (0, 1) else 2 * 2.0 * C * C * 6 * ns / 4 * L * 32 / 1e9
                tag = NAMES[mode] + (f" chv={chv}" if mode == 3 else "")
                row.append(f"{tag}: {ms.value * 1e3:7.1f} us ({gf_exec / ms.value:5.1f} TF exec)")
            print(f"C={C} k={k} d={d} epi={epi}:  " + "   ".join(row), flush=True)
