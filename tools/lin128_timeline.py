#!/usr/bin/env python
"""Timeline of one lin128_kernel launch (option kernel_dbg=32: wave 0 of every workgroup stamps the 100 MHz wall clock at its start,
loop start, loop end, after issuing its stores and after their acknowledgement, plus XCC_ID / HW_ID): how the rounds line up
on a CU, what a workgroup's prologue / main loop / epilogue take, and when slots are re-filled.
   python tools/lin128_timeline.py [cin cout [stagger]]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dissc_amd._lib import lib, check

cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (768, 3072)
stagger = int(sys.argv[3]) if len(sys.argv) > 3 else 0
one = len(sys.argv) > 4  # a 4th argument: ONE workgroup per CU (needs the dbg-32 instance to honour DISSC_LIN128_LDS)
B, T = 32, 499
ms = ctypes.c_float()
check(lib.dissc_set_option(b"lin128", 2), "set")
check(lib.dissc_conv_bench(B, cin, cout, 1, 1, T, 0, 1500, 1, ctypes.byref(ms)), "warm")
print(f"{cin}->{cout}: {ms.value * 1e3:.1f} us per launch (sustained, no stamps)")
check(lib.dissc_set_option(b"kernel_dbg", 32), "set")
os.environ["DISSC_TIMELINE"] = "/tmp/lin128_tl.bin"
check(lib.dissc_conv_bench(B, cin, cout, 1, 1, T, 0, 200, 1, ctypes.byref(ms)), "stamped")
print(f"with stamps: {ms.value * 1e3:.1f} us per launch")
raw = np.fromfile("/tmp/lin128_tl.bin", dtype=np.uint64).reshape(-1, 8)
n = int((raw[:, 0] != 0).sum())
raw = raw[:n]
t0 = raw[:, 0].min()
st, lp, le, si, ak = [(raw[:, i].astype(np.int64) - int(t0)) / 100.0 for i in (0, 1, 2, 3, 5)]  # us
hw = raw[:, 4]
cu = ((hw >> 32) << 12) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15)
print(f"{n} workgroups stamped; launch span {ak.max():.1f} us")
print(f"prologue (start -> loop): median {np.median(lp - st):.2f} us, p90 {np.percentile(lp - st, 90):.2f}")
print(f"main loop:                median {np.median(le - lp):.2f} us, p10 {np.percentile(le - lp, 10):.2f}, p90 {np.percentile(le - lp, 90):.2f}")
print(f"epilogue issue:           median {np.median(si - le):.2f} us, p90 {np.percentile(si - le, 90):.2f}")
print(f"store acknowledgement:    median {np.median(ak - si):.2f} us, p90 {np.percentile(ak - si, 90):.2f}")
order = np.argsort(st)
rounds = np.array_split(order, max(1, round(n / 512)))
for i, r in enumerate(rounds):
    print(f"round {i}: starts {st[r].min():.1f} .. {st[r].max():.1f} us (median {np.median(st[r]):.1f}), ends {ak[r].min():.1f} .. {ak[r].max():.1f} "
          f"(median {np.median(ak[r]):.1f}); loop {np.median((le - lp)[r]):.1f} us")
for c in list(dict.fromkeys(cu.tolist()))[:3]:
    ids = np.nonzero(cu == c)[0]
    ids = ids[np.argsort(st[ids])]
    print(f"CU {c:05x}: " + "; ".join(f"wg {i}: {st[i]:.1f} [{lp[i]:.1f} .. {le[i]:.1f}] {si[i]:.1f} / {ak[i]:.1f}" for i in ids))
