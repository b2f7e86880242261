#!/usr/bin/env python
"""one stride-2 feature conv shape under one conv2s128 variant (for rocprofv3 --pmc runs): python tools/one_s2.py variant [L_in [iters]]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dissc_amd._lib import lib, check
v = int(sys.argv[1]); L = int(sys.argv[2]) if len(sys.argv) > 2 else 31999; it = int(sys.argv[3]) if len(sys.argv) > 3 else 4
check(lib.dissc_set_option(b"conv2s128", v), "set")
ms = ctypes.c_float()
check(lib.dissc_conv_s2_bench(32, 512, L, 0, it, ctypes.byref(ms)), "bench")
print(f"conv2s128={v} L={L}: {ms.value * 1e3:.1f} us")
