#!/usr/bin/env python
"""Gate of conv2s128_kernel (csrc/lin_gemm.hip): HuBERT's stride-2, k = 3 feature convs (conv1 .. conv4, B = 32 x 10 s) through
dissc_conv_s2_bench under conv2s128 = 0 (conv_mfma32_kernel, 256 x 64 tiles) / 1 (16 channels per barrier) / 3 (32), sustained, alternating -- and the error of both
kernels against a float64 convolution (ragged lengths, NaN-poisoned padding; another K order: not bit-identical, same error level).
   python tools/conv2s128_gate.py [variants ...]      (default: 0 1 3)"""
import ctypes, os, sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dissc_amd._lib import lib, check

vals = [int(v) for v in sys.argv[1:]] or [0, 1, 3]


def conv(x, w, bias, lengths, v, act):
    check(lib.dissc_set_option(b"conv2s128", v), "set")
    Bn, cin, L = x.shape
    cout = w.shape[0]
    Lo = (L - 3) // 2 + 1
    ldx, ldo = (L + 3) // 4 * 4, (Lo + 3) // 4 * 4
    xd = torch.full((Bn, cin, ldx), float("nan"), device="cuda")
    for i, n in enumerate(lengths):
        xd[i, :, :n] = x[i, :, :n].cuda()
    yd = torch.full((Bn, cout, ldo), -7.0, device="cuda")
    ln = torch.as_tensor(lengths, dtype=torch.int32).cuda()
    wc, bc = w.contiguous(), bias.contiguous()
    check(lib.dissc_conv1d_s2(xd.data_ptr(), wc.data_ptr(), bc.data_ptr(), yd.data_ptr(), ln.data_ptr(), Bn, cin, cout, 3, ldx, ldo,
                              L, act, 0, None), "conv1d_s2")
    torch.cuda.synchronize()
    return yd.cpu()


def bench(L, v, iters):
    check(lib.dissc_set_option(b"conv2s128", v), "set")
    ms = ctypes.c_float()
    check(lib.dissc_conv_s2_bench(32, 512, L, 0, iters, ctypes.byref(ms)), "bench")
    return ms.value * 1e3


if __name__ == "__main__":
    rs = np.random.RandomState(0)
    for cin, cout, L, lengths, act in [(512, 512, 1031, [1031, 3, 4, 259, 516, 777], 1), (64, 256, 300, [300, 257], 0),
                                       (512, 512, 31999, [31999], 1)]:
        x = torch.from_numpy(rs.standard_normal((len(lengths), cin, L)).astype(np.float32))
        w = torch.from_numpy((rs.standard_normal((cout, cin, 3)) / np.sqrt(3 * cin)).astype(np.float32))
        bias = torch.from_numpy(rs.standard_normal(cout).astype(np.float32))
        ys = {v: conv(x, w, bias, lengths, v, act) for v in vals}
        for i, n in enumerate(lengths):
            no = (n - 3) // 2 + 1
            ref = F.conv1d(x[i:i + 1, :, :n].double(), w.double(), bias.double(), stride=2)
            if act:
                ref = F.gelu(ref)
            assert ref.shape[-1] == no
            row = []
            for v in vals:
                y = ys[v][i]
                assert bool((y[:, no:] == -7.0).all()), "padding written"
                e = (y[:, :no].double() - ref[0])
                row.append(f"conv2s128={v}: max {float(e.abs().max()):.2e} rms {float(e.pow(2).mean().sqrt()):.2e}")
                assert float(e.abs().max()) <= 2e-5, (v, i)
            print(f"{cin}->{cout} L={L} n={n} -> {no}: " + " | ".join(row), flush=True)
    bench(31999, 0, 100)  # warm the chip up
    tot = {v: 0.0 for v in vals}
    for name, lin, iters in [("conv1", 31999, 100), ("conv2", 15999, 200), ("conv3", 7999, 400), ("conv4", 3999, 800)]:
        lo = (lin - 3) // 2 + 1
        gf = 2.0 * 512 * 512 * 3 * lo * 32 / 1e9
        best = {v: 1e9 for v in vals}
        for _ in range(2):
            for v in vals:
                best[v] = min(best[v], bench(lin, v, iters))
        for v in vals:
            tot[v] += best[v]
        print(f"{name} L_in {lin:5d}: " + " | ".join(f"conv2s128={v}: {best[v]:7.1f} us {gf / best[v] * 1e3:6.1f} TFLOP/s" for v in vals),
              flush=True)
    print("conv1..4: " + " | ".join(f"conv2s128={v}: {tot[v] / 1e3:.3f} ms" for v in vals))
    if os.environ.get("CONV2S_TL"):
        os.environ["DISSC_TIMELINE"] = "/tmp/c2s_tl.bin"
        check(lib.dissc_set_option(b"kernel_dbg", 32), "set")
        us = bench(31999, 1, 50)
        check(lib.dissc_set_option(b"kernel_dbg", 0), "set")
        raw = np.fromfile("/tmp/c2s_tl.bin", dtype=np.uint64).reshape(-1, 8)
        raw = raw[raw[:, 2] != 0]
        t0 = raw[:, 0].min()
        st, lp, le, si, ak = [(raw[:, i].astype(np.int64) - int(t0)) / 100.0 for i in (0, 1, 2, 3, 5)]
        # residency per CU: how much of the time are both slots of a CU occupied?  (window: until the first CU runs out of stamped ids)
        hw = raw[:, 4]
        cu = ((hw >> 32) << 12) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15)
        xcc = (hw >> 32).astype(np.int64)
        tend = min(st[cu == c].max() for c in np.unique(cu))
        occ = np.zeros(3)
        for c in np.unique(cu):
            ev = sorted([(t, 1) for t in st[cu == c]] + [(t, -1) for t in ak[cu == c]])
            n, tprev = 0, 0.0
            for t, d in ev:
                t = min(t, tend)
                occ[min(n, 2)] += t - tprev
                tprev, n = t, n + d
                if t >= tend:
                    break
        print(f"residency over the first {tend:.0f} us, all CUs: two workgroups {occ[2] / occ.sum():.3f}, one {occ[1] / occ.sum():.3f}, none "
              f"{occ[0] / occ.sum():.3f}")
        for x in range(8):
            m = xcc == x
            print(f"  XCD {x}: {int(m.sum())} stamped workgroups, last stamped start {st[m].max():.0f} us, median loop {np.median((le - lp)[m]):.1f} us")
        print(f"conv1 with stamps {us:.0f} us; first {len(raw)} workgroup ids: prologue median {np.median(lp - st):.2f} us, main loop median "
              f"{np.median(le - lp):.1f} (p10 {np.percentile(le - lp, 10):.1f}, p90 {np.percentile(le - lp, 90):.1f}; two waves sharing a "
              f"SIMD at full rate: 329), epilogue issue {np.median(si - le):.2f}, acknowledged {np.median(ak - si):.2f}")
