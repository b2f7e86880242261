#!/usr/bin/env python
"""Does the 256 MiB Infinity Cache help the HBM-bound layers when the batch is cut so that a layer's tensors fit it?
One conv layer shape through dissc_conv_bench at B = 32 / 16 / 8 / 4: time per utterance.  If time / B falls at small B
(tensors of C x L x B x 4 bytes under ~100 MB), running the wide-tensor stages in sub-batches would pay.
    python tools/mall_probe.py"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dissc_amd._lib import lib, check  # noqa: E402

ms = ctypes.c_float()
for (C, L, k, d, epi) in ((64, 40000, 3, 1, 0), (64, 40000, 3, 1, 1), (64, 40000, 7, 1, 1), (32, 80000, 3, 1, 1),
                          (16, 160000, 3, 1, 1), (16, 160000, 7, 5, 3), (128, 10000, 3, 1, 1)):
    row = []
    for B in (32, 16, 8, 4):
        check(lib.dissc_conv_bench(B, C, C, k, d, L, epi, 30, 0, ctypes.byref(ms)), "conv_bench")
        row.append((B, ms.value))
    mb = C * L * 4 / 1e6
    print(f"C={C} L={L} k={k} d={d} epi={epi} ({mb:.1f} MB per utterance and tensor): " +
          "  ".join(f"B={B}: {t * 1e3:.0f} us = {t * 1e3 / B:.2f} us/utt" for B, t in row), flush=True)
