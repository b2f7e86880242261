import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import dissc_amd
from dissc_amd._lib import check
L = dissc_amd.lib
ms = ctypes.c_float()
for k in (7, 11):
    for d in (1, 3, 5):
        for epi in (0, 1, 3):
            row = []
            check(L.dissc_conv_bench(32, 64, 64, k, d, 40000, epi, 20, 2, ctypes.byref(ms)), "b"); row.append(f"F43 {ms.value*1e3:5.0f}")
            for wide in (1, 0, 2):
                check(L.dissc_set_option(b"wino8_c64_wide", wide), "o")
                check(L.dissc_conv_bench(32, 64, 64, k, d, 40000, epi, 20, 4, ctypes.byref(ms)), "b"); row.append(f"F63 mode{wide} {ms.value*1e3:5.0f}")
            print(f"C64 k{k} d{d} epi{epi}: " + "  ".join(row), flush=True)
L.dissc_set_option(b"wino8_c64_wide", 1)
