#!/usr/bin/env python
"""linear (1x1, no activation on load) layer shapes under the "lin_dma" variants (0 = register staging, 1 = LDS-DMA with
64 channels per barrier, 2 = with 32): python tools/lin_ab.py 0 1 2"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dissc_amd._lib import lib, check
vals = [int(v) for v in sys.argv[1:]] or [0, 1, 2]
for C, L in [(768, 2048), (768, 500), (3072, 512), (1536, 1024)]:
    out = []
    for v in vals * 2:
        check(lib.dissc_set_option(b"lin_dma", v), "set")
        ms = ctypes.c_float()
        check(lib.dissc_conv_bench(32, C, C, 1, 1, L, 0, 20, 1, ctypes.byref(ms)), "bench")  # flags bit 0: no activation on load
        out.append(f"lin_dma={v}: {ms.value*1e3:6.0f} us {2.0*C*C*L*32/ms.value/1e9:6.1f} TF")
    print(f"C={C} L={L}: " + " | ".join(out), flush=True)
