#!/usr/bin/env python
"""Run one conv shape repeatedly (for rocprofv3 --pmc passes) and print the MFMA ceiling.
usage: one_conv.py C L k d epi iters"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dissc_amd._lib import lib, check  # noqa: E402

C, L, k, d, epi, iters = (int(x) for x in sys.argv[1:7])
tf = ctypes.c_float()
if "--peak" in sys.argv:
    check(lib.dissc_mfma_peak(20000, ctypes.byref(tf)), "mfma_peak")
    print(f"fp32 MFMA sustained: {tf.value:.1f} TFLOP/s")
ms = ctypes.c_float()
check(lib.dissc_conv_bench(32, C, C, k, d, L, epi, iters, 0, ctypes.byref(ms)), "conv_bench")
print(f"C={C} L={L} k={k} d={d} epi={epi}: {ms.value:.3f} ms  {2.0*C*C*k*L*32/ms.value/1e9:.1f} TFLOP/s")
