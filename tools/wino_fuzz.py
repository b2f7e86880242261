#!/usr/bin/env python
"""Randomised cross-check of conv_wino_kernel against the direct kernel through dissc_conv1d: random (C, k, d), batch
sizes, row lengths and ragged utterance lengths (incl. 1, tile boundaries, multiples of nothing), NaN-poisoned padding,
sentinel-filled outputs.  Every transform form (wino_sv 0 / 1) and tile size (small_grid 0 / 1) must give the SAME bits;
the result must agree with the direct kernel to fp32 rounding and nothing may be written beyond an utterance.

    python tools/wino_fuzz.py [cases] [seed]
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dissc_amd
from dissc_amd._lib import check

L = dissc_amd.lib
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def run(x, w, b, lengths, k, d, ld):
    B, C, _ = x.shape
    y = torch.full((B, C, ld), -7.0, device="cuda")
    check(L.dissc_conv1d(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), lengths.data_ptr(), B, C, C, k, d, ld, ld,
                         int(lengths.max()), ctypes.c_float(0.1), None), "conv")
    torch.cuda.synchronize()
    return y


worst = 0.0
for case in range(n_cases):
    C = int(rs.choice([64, 128, 256]))
    k = int(rs.choice([3, 7, 11]))
    d = int(rs.choice([1, 3, 5]))
    B = int(rs.choice([1, 2, 3, 5, 9, 17, 40, 70, 130]))  # > 64: the tile search walks the utterances 64 at a time
    Lmax = int(rs.choice([1, 5, 63, 240, 256, 257, 511, 777, 1024, 2500, 4099]))
    lengths = rs.randint(1, Lmax + 1, size=B).astype(np.int32)
    lengths[rs.randint(B)] = Lmax
    ld = (Lmax + 3) // 4 * 4 + 4 * int(rs.randint(0, 3))
    x = torch.from_numpy(rs.standard_normal((B, C, ld)).astype(np.float32)).cuda()
    for i, n in enumerate(lengths):
        x[i, :, int(n):] = float("nan")
    w = torch.from_numpy((rs.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32))
    bias = torch.from_numpy(rs.standard_normal(C).astype(np.float32))
    ln = torch.from_numpy(lengths).cuda()
    outs = {}
    try:
        check(L.dissc_set_option(b"wino", 0), "opt")
        direct = run(x, w, bias, ln, k, d, ld)
        check(L.dissc_set_option(b"wino", 2), "opt")
        for sv in (0, 1):
            for sg in (0, 1):
                check(L.dissc_set_option(b"wino_sv", sv), "opt")
                check(L.dissc_set_option(b"small_grid", sg), "opt")
                outs[(sv, sg)] = run(x, w, bias, ln, k, d, ld)
    finally:
        L.dissc_set_option(b"wino", 1)
        L.dissc_set_option(b"wino_sv", 1)
        L.dissc_set_option(b"small_grid", 1)
    ref = outs[(1, 1)]
    for key, y in outs.items():
        assert torch.equal(y, ref), f"case {case}: form {key} differs (C={C} k={k} d={d} B={B} Lmax={Lmax})"
    err = 0.0
    for i, n in enumerate(lengths):
        n = int(n)
        assert (ref[i, :, n:] == -7.0).all(), f"case {case}: wrote beyond utterance {i}"
        assert torch.isfinite(ref[i, :, :n]).all(), f"case {case}: non-finite output"
        err = max(err, float((ref[i, :, :n] - direct[i, :, :n]).abs().max()))
    worst = max(worst, err)
    assert err <= 3e-5, f"case {case}: max |wino - direct| = {err} (C={C} k={k} d={d} B={B} Lmax={Lmax})"
    print(f"case {case:3d}: C={C:3d} k={k:2d} d={d} B={B:2d} Lmax={Lmax:4d} ld={ld:4d}  max |wino - direct| {err:.2e}", flush=True)
print(f"{n_cases} cases ok, worst difference to the direct kernel {worst:.2e}")
