"""Gate of the polyphase Toom-Cook feature convs (verdict round 4, task 1): per-launch time of HuBERT's conv1..conv4 shapes
(B = 32 x 10 s) in both forms, and the knock-outs of the new kernel.  Usage: python tools/s2tc_gate.py [iters]"""
import ctypes
import sys

import torch  # noqa: F401
from dissc_amd import _lib as dissc_amd

L = dissc_amd.lib
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def bench(B, C, Lin, form):
    ms = ctypes.c_float()
    dissc_amd.check(L.dissc_conv_s2_bench(B, C, Lin, form, iters, ctypes.byref(ms)), "conv_s2_bench")
    return ms.value


shapes = [("conv1", 31999), ("conv2", 15999), ("conv3", 7999), ("conv4", 3999)]
tot = [0.0, 0.0]
for name, lin in shapes:
    lo = (lin - 3) // 2 + 1
    gf = 2.0 * 512 * 512 * 3 * lo * 32 / 1e9
    t0 = bench(32, 512, lin, 0)
    t1 = bench(32, 512, lin, 1)
    tot[0] += t0
    tot[1] += t1
    print(f"{name}: L_in {lin} -> {lo}: direct {t0 * 1e3:8.1f} us ({gf / t0:6.1f} TFLOP/s)   toom-cook {t1 * 1e3:8.1f} us "
          f"({gf / t1:6.1f} algorithmic, {gf * 15 / 21 / t1:6.1f} executed TFLOP/s)   x{t0 / t1:.3f}", flush=True)
print(f"conv1..4: direct {tot[0]:.3f} ms, toom-cook {tot[1]:.3f} ms")
for mode in (0, 1):
    L.dissc_set_option(b"s2tc_xmode", mode)
    print(f"conv1 xmode {mode}: {bench(32, 512, 31999, 1) * 1e3:8.1f} us", flush=True)
L.dissc_set_option(b"s2tc_xmode", 0)
for dbg, what in ((1, "no transform"), (2, "no MFMAs"), (4, "no epilogue"), (8, "no window staging"), (9, "no staging, no transform"),
                  (13, "MFMA loop with A loads and B reads only"), (29, "... without the A loads"), (45, "... without the B reads"),
                  (61, "MFMAs + barriers only"), (7, "skeleton + staging"), (15, "skeleton")):
    L.dissc_set_option(b"s2tc_dbg", dbg)
    print(f"conv1 knock-out {dbg:2d} ({what}): {bench(32, 512, 31999, 1) * 1e3:8.1f} us", flush=True)
L.dissc_set_option(b"s2tc_dbg", 0)
