"""GPU parity of the polyphase Toom-Cook form of HuBERT's stride-2, k = 3 feature convs (dissc_amd/csrc/conv_s2tc.hip)
against a float64 ``F.conv1d(stride=2)`` of the same op and against the direct implicit-GEMM form it replaces
(reference: fairseq ConvFeatureExtractionModel behind data/encode.py:21-22,32; HF modeling_hubert.py:154-213).

Tolerances (written here, floating point): both forms within 2e-5 of the float64 result relative to the output's rms
(K = 3 x Cin fp32 products per output); the transform-domain form's rms error at most 3x the direct form's on the
512 -> 512 layer shape (the numpy model of F(7,2) on these points says 2.7x a blocked-fp32 conv; measured values are printed);
utterances are independent of what lies behind their end (NaN-poisoned padding)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from conftest import is_experimental_build
    from dissc_amd import _lib
    if not is_experimental_build():
        # the form's gate failed (conv1 5.95 ms against <= 5.3, rms 2.35x against <= 2x: profiles/r05/s2tc_gate.txt): the default
        # library does not carry the kernel; what the default build must do is refuse it loudly
        x = torch.zeros(1, 16, 64, device="cuda:0")
        w = torch.zeros(64, 16, 3)
        y = torch.zeros(1, 64, 32, device="cuda:0")
        rc = _lib.lib.dissc_conv1d_s2(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), None,
                                      ctypes.c_void_p(y.data_ptr()), None, 1, 16, 64, 3, 64, 32, 64, 0, 1, None)
        assert rc != 0 and b"DISSC_EXPERIMENTAL" in _lib.lib.dissc_last_error()
        pytest.skip("conv_s2tc.hip is only in DISSC_EXPERIMENTAL=1 builds (its gate failed)")
    return _lib


def _run(lib, x, w, lens, form, act=0, bias=None, poison=True):
    """x [B,Cin,L] cpu f32, lens list -> y [B,Cout,Lo_max] cpu (NaN where nothing was stored)"""
    B, Cin, L = x.shape
    Cout, k = w.shape[0], w.shape[2]
    ldx = (L + 3) // 4 * 4
    Lo = (L - k) // 2 + 1
    ldo = (Lo + 3) // 4 * 4
    xd = torch.full((B, Cin, ldx), float("nan") if poison else 0.0, device="cuda:0")
    for b in range(B):
        xd[b, :, :lens[b]] = x[b, :, :lens[b]].cuda()
    yd = torch.full((B, Cout, ldo), float("nan"), device="cuda:0")
    ld = torch.tensor(lens, dtype=torch.int32, device="cuda:0")
    wc = w.contiguous()
    bc = bias.contiguous() if bias is not None else None
    lib.check(lib.lib.dissc_conv1d_s2(ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(wc.data_ptr()),
                                      ctypes.c_void_p(bc.data_ptr()) if bc is not None else None,
                                      ctypes.c_void_p(yd.data_ptr()), ctypes.c_void_p(ld.data_ptr()), B, Cin, Cout, k, ldx, ldo,
                                      L, act, form, None), "dissc_conv1d_s2")
    return yd.cpu()


def _ref(x, w, n, act=0, bias=None):
    y = F.conv1d(x[:, :n].double()[None], w.double(), bias.double() if bias is not None else None, stride=2)[0]
    return F.gelu(y) if act else y


@pytest.mark.parametrize("Cin,Cout,L,lens", [
    (64, 64, 2000, [2000, 1999, 901, 3, 2]),       # 2 samples: no output at all
    (32, 128, 1800, [1800, 1795, 897, 898, 899]),  # ends around a tile boundary (448 outputs = 897 samples)
    (16, 64, 40, [40, 17, 5]),                     # a single partial unit
    (512, 512, 3001, [3001, 2000]),                # the layer's own width
])
def test_s2tc_matches_float64_and_direct(lib, Cin, Cout, L, lens):
    g = torch.Generator().manual_seed(Cin * 7 + L)
    x = F.gelu(torch.randn(len(lens), Cin, L, generator=g))
    w = torch.randn(Cout, Cin, 3, generator=g) * (2.0 / (3 * Cin)) ** 0.5
    y1 = _run(lib, x, w, lens, 1, act=1)
    y0 = _run(lib, x, w, lens, 0, act=1) if Cout >= 256 else y1  # (the direct strided instance needs >= 256 output rows)
    for b, n in enumerate(lens):
        no = (n - 3) // 2 + 1 if n >= 3 else 0
        ref = _ref(x[b], w, n, act=1) if no > 0 else torch.zeros(Cout, 0, dtype=torch.float64)
        assert torch.isnan(y1[b, :, no:]).all(), "stored beyond the utterance's output length"
        if no == 0:
            continue
        got1, got0 = y1[b, :, :no].double(), y0[b, :, :no].double()
        assert torch.isfinite(got1).all()
        scale = float(ref.pow(2).mean().sqrt())
        e1 = float((got1 - ref).pow(2).mean().sqrt()) / scale
        e0 = float((got0 - ref).pow(2).mean().sqrt()) / scale
        m1 = float((got1 - ref).abs().max()) / scale
        print(f"Cin {Cin} Cout {Cout} len {n}: rel rms toom-cook {e1:.2e} direct {e0:.2e} (max {m1:.2e})")
        assert e1 <= 2e-5 and e0 <= 2e-5 and m1 <= 2e-4, (e1, e0, m1)


def test_s2tc_bias_and_no_activation(lib):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 333, generator=g)
    w = torch.randn(64, 32, 3, generator=g) * 0.2
    bias = torch.randn(64, generator=g)
    y = _run(lib, x, w, [333, 200], 1, act=0, bias=bias)
    for b, n in enumerate([333, 200]):
        ref = _ref(x[b], w, n, bias=bias)
        assert float((y[b, :, :ref.shape[1]].double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def test_s2tc_is_batch_independent(lib):
    """an utterance's outputs do not depend on its neighbours or on what lies behind its end: bit-identical alone vs batched"""
    g = torch.Generator().manual_seed(11)
    x = F.gelu(torch.randn(3, 64, 1500, generator=g))
    w = torch.randn(64, 64, 3, generator=g) * 0.1
    lens = [1500, 1234, 450]
    yb = _run(lib, x, w, lens, 1, act=1)
    for b, n in enumerate(lens):
        ya = _run(lib, x[b:b + 1, :, :n].contiguous(), w, [n], 1, act=1, poison=False)
        no = (n - 3) // 2 + 1
        assert torch.equal(ya[0, :, :no], yb[b, :, :no])


def test_s2tc_error_against_direct_form_on_the_layer_shape(lib):
    """the gate's accuracy half: 512 -> 512 channels, GELU-shaped inputs, rms error vs float64 of both forms"""
    g = torch.Generator().manual_seed(5)
    x = F.gelu(torch.randn(1, 512, 8001, generator=g))
    w = torch.randn(512, 512, 3, generator=g) * (2.0 / 1536) ** 0.5
    ref = _ref(x[0], w, 8001)
    e = []
    for form in (1, 0):
        y = _run(lib, x, w, [8001], form)[0, :, :ref.shape[1]].double()
        e.append(float((y - ref).pow(2).mean().sqrt()))
    print(f"512x512 k3 s2: rms error toom-cook {e[0]:.3e}, direct {e[1]:.3e}, ratio {e[0] / e[1]:.2f} (output rms {float(ref.pow(2).mean().sqrt()):.3f})")
    assert e[0] <= 3.0 * e[1], e
