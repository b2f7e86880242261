import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dissc_amd._lib import lib
lib.dissc_pipe_overlap.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
ms = (ctypes.c_float * 3)()
for mi, vi in ((20000, 40000), (20000, 80000), (40000, 40000)):
    assert lib.dissc_pipe_overlap(mi, vi, ms) == 0
    mf = 2.0 * 32 * 32 * 2 * 4 * mi * 4 * 512 / (ms[0] * 1e-3) / 1e12
    vf = 2.0 * 16 * vi * 256 * 512 / (ms[1] * 1e-3) / 1e12
    print(f"mfma alone {ms[0]:.2f} ms ({mf:.0f} TF)  valu alone {ms[1]:.2f} ms ({vf:.0f} TF)  together {ms[2]:.2f} ms  (sum {ms[0]+ms[1]:.2f}, max {max(ms[0],ms[1]):.2f})")
