#!/usr/bin/env python
"""GPU tuning sweep over generator conv shapes x tile configs of the 16x16x4 kernel
(diagnostics entry point dissc_conv_bench; run with DISSC_OPTIONS=mfma32=0 -- the 32x32x2
kernel's tile shapes are tuned with tools/option_sweep.py instead).  Prints a markdown table."""
import ctypes
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dissc_amd  # noqa: E402
from dissc_amd._lib import lib, check

B, T = 32, 500
STAGES = [(256, 5 * T), (128, 20 * T), (64, 80 * T), (32, 160 * T), (16, 320 * T)]
CFG_NAMES = {0: "256x64", 1: "128x128", 2: "64x256", 3: "32x512", 4: "16x512", 5: "32x256",
             6: "16x256", 7: "64x128", 8: "128x64", 9: "256x64w8"}
CANDS = {256: [0, 9], 128: [1, 8], 64: [2, 7], 32: [3, 5], 16: [4, 6]}


def run(C, L, k, d, epi, flags=0, cfg=None, iters=5):
    f = flags
    if cfg is not None:
        cls = {16: 0, 32: 1, 64: 2, 128: 3, 256: 4}[C]
        f |= 0x8000 | (cfg << 8) | (cls << 16)
    ms = ctypes.c_float()
    check(lib.dissc_conv_bench(B, C, C, k, d, L, epi, iters, f, ctypes.byref(ms)), "conv_bench")
    return ms.value


def main():
    quick = "--quick" in sys.argv
    print("| C | L | k | d | epi | cfg | ms | TFLOP/s |")
    print("|---|---|---|---|---|---|---|---|")
    for C, L in STAGES:
        for k, d in ((3, 1), (11, 5)) if quick else ((3, 1), (3, 5), (7, 3), (11, 1), (11, 5)):
            for epi in (0, 1):
                for cfg in CANDS[C]:
                    fl = 2.0 * C * C * k * L * B
                    ms = run(C, L, k, d, epi, 0, cfg)
                    print(f"| {C} | {L} | {k} | {d} | {epi} | {CFG_NAMES[cfg]} | {ms:.3f} | "
                          f"{fl/ms/1e9:.1f} |", flush=True)


if __name__ == "__main__":
    main()
