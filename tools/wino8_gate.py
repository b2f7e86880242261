#!/usr/bin/env python
"""Gate experiment for conv_wino8.hip (Toom-Cook F(6,3) on 8-wave workgroups, k = 7 / 11): per-layer error against a
float64 F.conv1d next to the F(4,3) form's and the direct kernel's, then timing against the F(4,3) form through
dissc_conv_bench at the generator's shapes (B = 32 x 10 s).
    python tools/wino8_gate.py [check] [time]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import dissc_amd
from dissc_amd._lib import check

L = dissc_amd.lib
dev = "cuda:0"
what = sys.argv[1:] or ["check", "time"]


def conv1d(x, w, b, lengths, d, slope, form):  # form: 0 direct, 4 F(4,3), 6 F(6,3), 5 F(5,4) (where there is an instance)
    assert L.dissc_set_option(b"wino", 2 if form == 4 else 1) == 0
    assert L.dissc_set_option(b"wino8", 2 if form in (5, 6) else 0) == 0
    assert L.dissc_set_option(b"wino8_r4", 2 if form == 5 else 0) == 0
    if form == 0:
        assert L.dissc_set_option(b"wino", 0) == 0
    B, C, ld = x.shape
    y = torch.full_like(x, float("nan"))
    check(L.dissc_conv1d(x.data_ptr(), w.contiguous().data_ptr(), b.data_ptr(), y.data_ptr(), lengths.data_ptr(), B, C,
                         w.shape[0], w.shape[2], d, ld, ld, int(lengths.max()), ctypes.c_float(slope), None), "conv1d")
    torch.cuda.synchronize()
    L.dissc_set_option(b"wino", 1)
    L.dissc_set_option(b"wino8", 1)  # (the defaults)
    L.dissc_set_option(b"wino8_r4", 1)
    return y


if "check" in what:
    torch.manual_seed(0)
    worst = 0.0
    for C in (64, 128, 256):
        for k in (7, 11):
            for d in (1, 3, 5):
                lens = [1000, 1, 7, 359, 360, 361, 613, 997]
                ld = 1000
                x = torch.rand(len(lens), C, ld, device=dev) * 2 - 1
                lengths = torch.tensor(lens, dtype=torch.int32, device=dev)
                for i, n in enumerate(lens):
                    x[i, :, n:] = float("nan")
                w = (torch.rand(C, C, k) * 2 - 1) * 0.025 * (256 / C) ** 0.5
                b = torch.rand(C) * 0.2 - 0.1
                errs = {}
                r4 = True
                for form in (0, 4, 6) + ((5,) if r4 else ()):
                    y = conv1d(x, w, b, lengths, d, 0.1, form)
                    e2 = n2 = 0.0
                    mx = 0.0
                    for i, n in enumerate(lens):
                        xi = x[i:i + 1, :, :n].double()
                        ref = F.conv1d(F.leaky_relu(xi, 0.1), w.double().to(dev), b.double().to(dev), padding=(k - 1) * d // 2, dilation=d)
                        got = y[i, :, :n].double()
                        assert torch.isfinite(got).all(), (C, k, d, form, i)
                        assert torch.isnan(y[i, :, n:]).all(), (C, k, d, form, i, "wrote beyond the utterance")
                        e = got - ref[0]
                        e2 += float((e ** 2).sum()); n2 += e.numel(); mx = max(mx, float(e.abs().max()))
                    errs[form] = ((e2 / n2) ** 0.5, mx)
                print(f"C={C} k={k} d={d}: rms / max error  direct {errs[0][0]:.2e} / {errs[0][1]:.2e}   F(4,3) {errs[4][0]:.2e} / {errs[4][1]:.2e}"
                      f"   F(6,3) {errs[6][0]:.2e} / {errs[6][1]:.2e}" + (f"   F(5,4) {errs[5][0]:.2e} / {errs[5][1]:.2e}" if r4 else ""), flush=True)
                for form in (6, 5) if r4 else (6,):
                    worst = max(worst, errs[form][1])
                    assert errs[form][0] <= 4.0 * errs[0][0] + 1e-8 and errs[form][1] <= 2e-5, form
    print(f"check ok, worst F(6,3) / F(5,4) max error {worst:.2e}")

if "time" in what:
    ms = ctypes.c_float()
    for C, Ls in ((256, 2500), (128, 10000), (64, 40000)):
        for k in (7, 11):
            for d in (1, 3, 5):
                row = []
                for epi in (0, 1, 3):
                    t = {}
                    r4 = True
                    for flag in (2, 4) + ((12,) if r4 else ()):
                        check(L.dissc_conv_bench(32, C, C, k, d, Ls, epi, 20, flag, ctypes.byref(ms)), "conv_bench")
                        t[flag] = ms.value * 1e3
                    row.append(f"epi {epi}: F(4,3) {t[2]:6.0f} us, F(6,3) {t[4]:6.0f} us ({t[2] / t[4]:.2f}x)" +
                               (f", F(5,4) {t[12]:6.0f} us ({t[2] / t[12]:.2f}x)" if r4 else ""))
                print(f"C={C} L={Ls} k={k} d={d}:  " + "   ".join(row), flush=True)
