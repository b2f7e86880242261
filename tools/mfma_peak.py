import ctypes, sys
sys.path.insert(0, '/root/repo')
from dissc_amd._lib import lib
tf = ctypes.c_float()
for it in (20000, -20000, 100000, -100000):
    lib.dissc_mfma_peak(it, ctypes.byref(tf)); print(it, round(tf.value, 1))
