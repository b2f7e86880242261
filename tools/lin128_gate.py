#!/usr/bin/env python
"""Gate of lin128_kernel (csrc/lin_gemm.hip; round 5 verdict, task 1): HuBERT's linear shapes at B = 32 x T = 499 through
dissc_conv_bench under lin128 = 0 (conv_mfma32_kernel, 256 x 64 tiles) and the lin128 variants, alternating, best of 3 --
and bit-identity of the two kernels' outputs through dissc_conv1d (ragged lengths, NaN-poisoned padding).
   python tools/lin128_gate.py [variants ...]      (default: 0 1)"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dissc_amd._lib import lib, check

vals = [int(v) for v in sys.argv[1:]] or [0, 1]
SHAPES = [("fc1 768->3072", 768, 3072, 0), ("fc2 3072->768 +res", 3072, 768, 1), ("qkv 768->2304", 768, 2304, 0),
          ("out 768->768 +res", 768, 768, 1), ("proj 512->768", 512, 768, 0)]
B, T = 32, 499


def bench(cin, cout, epi, v, iters=1000, flags=1):  # sustained: 20-launch runs read ~10 % slow (the chip is still ramping up)
    check(lib.dissc_set_option(b"lin128", v), "set")
    ms = ctypes.c_float()
    check(lib.dissc_conv_bench(B, cin, cout, 1, 1, T, epi, iters, flags, ctypes.byref(ms)), "bench")
    return ms.value * 1e3


def conv(x, w, bias, lengths, v):
    check(lib.dissc_set_option(b"lin128", v), "set")
    Bn, cin, L = x.shape
    cout = w.shape[0]
    ld = (L + 3) // 4 * 4
    xd = torch.full((Bn, cin, ld), float("nan"), device="cuda")
    for i, n in enumerate(lengths):
        xd[i, :, :n] = x[i, :, :n].cuda()
    yd = torch.full((Bn, cout, ld), -7.0, device="cuda")
    ln = torch.as_tensor(lengths, dtype=torch.int32).cuda()
    wc, bc = w.contiguous(), bias.contiguous()
    check(lib.dissc_conv1d(xd.data_ptr(), wc.data_ptr(), bc.data_ptr(), yd.data_ptr(), ln.data_ptr(), Bn, cin, cout, 1, 1, ld, ld,
                           L, ctypes.c_float(1.0), None), "conv1d")
    torch.cuda.synchronize()
    return yd.cpu()


if __name__ == "__main__":
    rs = np.random.RandomState(0)
    bench(768, 3072, 0, 0, iters=3000)  # warm the chip up
    for cin, cout, L, lengths in [(768, 3072, 499, [499, 1, 130, 257]), (3072, 768, 300, [300, 128, 129]), (512, 768, 131, [131, 4])]:
        x = torch.from_numpy(rs.standard_normal((len(lengths), cin, L)).astype(np.float32))
        w = torch.from_numpy((rs.standard_normal((cout, cin, 1)) / np.sqrt(cin)).astype(np.float32))
        bias = torch.from_numpy(rs.standard_normal(cout).astype(np.float32))
        y0 = conv(x, w, bias, lengths, 0)
        ref = torch.einsum("oc,bcl->bol", w[:, :, 0].double(), x.double()) + bias.double()[None, :, None]
        for v in vals[1:]:
            y1 = conv(x, w, bias, lengths, v)
            same = all(torch.equal(y0[i, :, :n], y1[i, :, :n]) for i, n in enumerate(lengths))
            untouched = all(bool((y1[i, :, n:] == -7.0).all()) for i, n in enumerate(lengths))
            err = max(float((y1[i, :, :n].double() - ref[i, :, :n]).abs().max()) for i, n in enumerate(lengths))
            print(f"{cin}->{cout} L={L} lengths={lengths}: lin128={v} bit-identical to lin128=0: {same}; padding untouched: {untouched}; "
                  f"max |err| vs float64 {err:.2e}", flush=True)
            assert same and untouched
    for name, cin, cout, epi in SHAPES:
        best = {v: 1e9 for v in vals}
        for _ in range(2):
            for v in vals:
                best[v] = min(best[v], bench(cin, cout, epi, v))
        gf = 2.0 * cin * cout * T * B / 1e9
        print(f"{name:20s} " + " | ".join(f"lin128={v}: {best[v]:6.1f} us {gf / best[v] * 1e3:6.1f} TFLOP/s" for v in vals), flush=True)
    if os.environ.get("LIN128_ZERO"):
        for v in vals:
            r, z = bench(768, 3072, 0, v), bench(768, 3072, 0, v, flags=1 | 0x20)
            print(f"fc1, lin128={v}: random operands {r:.1f} us, zero-filled operands {z:.1f} us ({r / z:.3f}x)", flush=True)
    if os.environ.get("LIN128_CLOCK"):
        # is the chip at its boost clock under a sustained dense fp32 GEMM?  (157.3 TFLOP/s assumes 2 400 MHz)
        import subprocess, threading
        samples = []

        def poll():
            for _ in range(6):
                out = subprocess.run("rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Socket'", shell=True,
                                     capture_output=True, text=True).stdout
                samples.append(" ".join(out.split()))
        for v in vals:
            samples.clear()
            th = threading.Thread(target=poll)
            th.start()
            us = bench(768, 3072, 0, v, iters=6000)
            th.join()
            print(f"sustained fc1, lin128={v}: {us:.1f} us per launch over 6000 launches")
            for smp in samples[1:5]:
                print("   ", smp[:200])
