"""The register-only Toom-Cook F(2,3) residual pairs of the default build (respair_f23.hip, respair16_f23.hip; k = 11 at C = 32 / 16) --
one residual pair y = x + conv_1(lrelu(conv_d(lrelu(x)))) (reference sr/models.py:34-41) per
launch -- through the C ABI (dissc_respair1d): against a float64 torch evaluation and the direct fused pair; ragged lengths, NaN
beyond every utterance, all epilogue modes.  (The F(4,3) pair kernel and the k = 3 instances: experimental/tests/test_gpu_pairw.py.)"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = [(32, 7, 1), (32, 7, 3), (32, 7, 5), (32, 11, 1), (32, 11, 3), (32, 11, 5), (64, 3, 1), (64, 3, 3), (64, 3, 5)]


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from dissc_amd import _lib
    return _lib


def _pair(lib, mode, x, w1, b1, w2, b2, lengths, k, d, epi=1, acc=None, slope=0.1, div=3.0):
    B, C, ld = x.shape
    y = torch.full_like(x, -7.0)
    ln = torch.as_tensor(lengths, dtype=torch.int32, device=DEV)
    a = None if acc is None else acc.clone()
    lib.check(lib.lib.dissc_respair1d(x.data_ptr(), w1.contiguous().data_ptr(), b1.data_ptr(), w2.contiguous().data_ptr(),
                                      b2.data_ptr(), y.data_ptr(), None if a is None else a.data_ptr(), ln.data_ptr(), B, C, k, d,
                                      ld, int(max(lengths)), ctypes.c_float(slope), epi, ctypes.c_float(div), mode, None),
              f"dissc_respair1d mode {mode}")
    return y if epi == 1 else a


def _reference(x, w1, b1, w2, b2, lengths, k, d, slope=0.1):
    """float64, one utterance at a time on its own samples (the reference runs B = 1: zero "same" padding at every layer)"""
    out = torch.zeros_like(x, dtype=torch.float64)
    for i, n in enumerate(lengths):
        xi = x[i:i + 1, :, :n].double()
        t = F.conv1d(F.leaky_relu(xi, slope), w1.double().to(x.device), b1.double().to(x.device), padding=(k - 1) * d // 2, dilation=d)
        y = F.conv1d(F.leaky_relu(t, slope), w2.double().to(x.device), b2.double().to(x.device), padding=(k - 1) // 2)
        out[i, :, :n] = xi[0] + y[0]
    return out


def _data(C, k, lengths, ld, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(len(lengths), C, ld, generator=g) * 2 - 1).to(DEV)
    for i, n in enumerate(lengths):
        x[i, :, n:] = float("nan")  # never read
    sc = 0.9 / (C * k) ** 0.5
    w1 = (torch.rand(C, C, k, generator=g) * 2 - 1) * sc
    w2 = (torch.rand(C, C, k, generator=g) * 2 - 1) * sc
    b1 = (torch.rand(C, generator=g) * 2 - 1) * 0.1
    b2 = (torch.rand(C, generator=g) * 2 - 1) * 0.1
    return x, w1, b1, w2, b2


F23_DEFAULT = 3  # the "pair_f23" mask the library ships with (bit 0: C = 32, bit 1: C = 16)


@pytest.fixture
def f23(lib):
    """mode 3 of dissc_respair1d builds the register-only F(2,3) forms (respair_f23.hip, respair16_f23.hip) for k = 11"""
    assert lib.lib.dissc_set_option(b"pair_f23", 15) == 0  # (every shape with an instance)
    yield
    lib.lib.dissc_set_option(b"pair_f23", F23_DEFAULT)


@pytest.mark.parametrize("k", [11])
@pytest.mark.parametrize("C,d", [(32, 1), (32, 3), (32, 5), (16, 1), (16, 3), (16, 5)])
def test_register_only_f23_pair_matches_float64_and_the_direct_pair(lib, f23, C, d, k):
    """respair32_f23_kernel / respair16_f23_kernel (k = 11: tiles of 500 / 492 / 468 outputs; k = 3: 508): ragged lengths around
    the tile edges, NaN beyond every utterance, against float64 and the direct pair; batch independence; the MRF modes"""
    lengths = [2000, 1, 7, 255, 467, 468, 469, 491, 492, 493, 499, 500, 501, 507, 508, 509, 1023, 1999, 12]
    ld = 2000
    x, w1, b1, w2, b2 = _data(C, k, lengths, ld, seed=900 + d + k)
    ref = _reference(x, w1, b1, w2, b2, lengths, k, d)
    y3 = _pair(lib, 3, x, w1, b1, w2, b2, lengths, k, d)
    y1 = _pair(lib, 1, x, w1, b1, w2, b2, lengths, k, d)  # the direct fused pair
    worst3 = worst1 = 0.0
    for i, n in enumerate(lengths):
        assert torch.isfinite(y3[i, :, :n]).all()
        assert (y3[i, :, n:] == -7.0).all(), f"utterance {i}: wrote beyond its {n} samples"
        worst3 = max(worst3, (y3[i, :, :n].double() - ref[i, :, :n]).abs().max().item())
        worst1 = max(worst1, (y1[i, :, :n].double() - ref[i, :, :n]).abs().max().item())
    r3 = float(((y3[0, :, :2000].double() - ref[0]) ** 2).mean().sqrt())
    r1 = float(((y1[0, :, :2000].double() - ref[0]) ** 2).mean().sqrt())
    print(f"C={C} k={k} d={d}: F(2,3) pair max err {worst3:.2e} rms {r3:.2e}; direct pair {worst1:.2e} / {r1:.2e}")
    assert not torch.equal(y3[0], y1[0])  # (the transform-domain kernel really ran)
    assert worst3 <= 1e-5 and r3 <= max(3.0 * r1, 1e-6)
    for i in (5, 16):
        one = _pair(lib, 3, x[i:i + 1].clone(), w1, b1, w2, b2, lengths[i:i + 1], k, d)
        assert torch.equal(one[0, :, :lengths[i]], y3[i, :, :lengths[i]])
    acc0 = torch.rand(len(lengths), C, ld, device=DEV)
    for epi in (2, 3, 4):
        a = _pair(lib, 3, x, w1, b1, w2, b2, lengths, k, d, epi=epi, acc=acc0)
        for i, n in enumerate(lengths):
            want = y3[i, :, :n] if epi == 2 else acc0[i, :, :n] + y3[i, :, :n]
            if epi == 4:
                want = (want.cpu() / 3.0).to(DEV)
            assert torch.equal(a[i, :, :n], want), (epi, i)
            assert torch.equal(a[i, :, n:], acc0[i, :, n:])

