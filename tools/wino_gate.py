#!/usr/bin/env python
"""Gate experiment for the Toom-Cook F(4,3) conv kernel (conv_wino.hip): per-layer error against F.conv1d (and the
direct kernel's error beside it), then direct vs transform-domain timing through dissc_conv_bench.
    python tools/wino_gate.py [check] [time]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
import dissc_amd
from dissc_amd._lib import check

L = dissc_amd.lib
dev = "cuda:0"
what = sys.argv[1:] or ["check", "time"]


def conv1d(x, w, b, lengths, d, slope, wino):
    assert L.dissc_set_option(b"wino", 2 if wino else 0) == 0
    B, C, ld = x.shape
    y = torch.full_like(x, float("nan"))
    Lmax = int(lengths.max())
    check(L.dissc_conv1d(x.data_ptr(), w.contiguous().data_ptr(), b.data_ptr(), y.data_ptr(), lengths.data_ptr(), B, C,
                         w.shape[0], w.shape[2], d, ld, ld, Lmax, ctypes.c_float(slope), None), "conv1d")
    torch.cuda.synchronize()
    return y


if "check" in what:
    torch.manual_seed(0)
    worst = 0.0
    for C in (64, 128, 256):
        for k in (3, 7, 11):
            for d in (1, 3, 5):
                lens = [1000, 1, 7, 255, 256, 257, 613]
                ld = 1000
                x = torch.rand(len(lens), C, ld, device=dev) * 2 - 1
                w = (torch.rand(C, C, k) * 2 - 1) * 0.025 * (256 / C) ** 0.5
                b = torch.randn(C) * 0.1
                lengths = torch.tensor(lens, dtype=torch.int32, device=dev)
                out = {}
                for wino in (0, 1):
                    out[wino] = conv1d(x, w, b, lengths, d, 0.1, wino)
                emax = {0: 0.0, 1: 0.0}
                erms = {0: 0.0, 1: 0.0}
                for i, n in enumerate(lens):
                    xa = F.leaky_relu(x[i:i + 1, :, :n].double(), 0.1)
                    ref = F.conv1d(xa, w.double().to(dev), b.double().to(dev), padding=(k - 1) * d // 2, dilation=d)[0]
                    ref32 = F.conv1d(xa.float(), w.to(dev), b.to(dev), padding=(k - 1) * d // 2, dilation=d)[0]
                    for wino in (0, 1):
                        got = out[wino][i, :, :n].double()
                        assert torch.isfinite(got).all(), (C, k, d, n, wino)
                        assert torch.isnan(out[wino][i, :, n:]).all(), "wrote beyond the utterance"
                        e = (got - ref).abs()
                        emax[wino] = max(emax[wino], float(e.max()))
                        erms[wino] = max(erms[wino], float((e ** 2).mean().sqrt()))
                    e32 = float((ref32.double() - ref).abs().max())
                worst = max(worst, emax[1])
                print(f"C={C:3d} k={k:2d} d={d}: direct max {emax[0]:.2e} rms {erms[0]:.2e} | wino max {emax[1]:.2e} rms "
                      f"{erms[1]:.2e} | torch fp32 max {e32:.2e} (ref rms {float(ref.std()):.2f})", flush=True)
    print("worst wino max error", worst)
    assert worst < 2e-5

if "time" in what:
    shapes = [(256, 11, 1, 2500), (256, 7, 1, 2500), (256, 3, 1, 2500), (256, 11, 5, 2500), (256, 3, 5, 2500), (128, 11, 1, 10000),
              (128, 7, 3, 10000), (128, 3, 1, 10000), (64, 11, 1, 40000), (64, 7, 1, 40000), (64, 3, 1, 40000), (64, 11, 5, 40000),
              (64, 7, 3, 40000), (64, 3, 5, 40000)]
    Bb = int(os.environ.get("WINO_B", "32"))
    for C, k, d, Ln in shapes:
        res = []
        for epi in (0, 1):
            for flags in (0, 2):
                ms = ctypes.c_float()
                best = 1e9
                for rep in range(2):
                    check(L.dissc_conv_bench(Bb, C, C, k, d, Ln, epi, 20, flags, ctypes.byref(ms)), "bench")
                    best = min(best, ms.value)
                res.append(best)
        gf = 2 * C * C * k * Ln * Bb / 1e9
        ns = (k + 2) // 3
        gfx = 2 * C * C * 6 * ns / 4 * Ln * Bb / 1e9
        print(f"C{C} k{k} d{d}: store direct {res[0]*1e3:7.1f} us ({gf/res[0]:6.1f} TF) wino {res[1]*1e3:7.1f} us (alg {gf/res[1]:6.1f} TF, "
              f"executed {gfx/res[1]:6.1f} TF) x{res[0]/res[1]:.2f} | +res direct {res[2]*1e3:7.1f} wino {res[3]*1e3:7.1f} x{res[2]/res[3]:.2f}",
              flush=True)
