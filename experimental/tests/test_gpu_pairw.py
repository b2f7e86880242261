"""respair_wino.hip -- one residual pair y = x + conv_1(lrelu(conv_d(lrelu(x)))) (reference sr/models.py:34-41) per
launch with both convs in the Toom-Cook F(4,3) transform domain -- through the C ABI (dissc_respair1d): against a
float64 torch evaluation, against the two-launch transform-domain path (bit-identical where conv_wino has an
instance) and against the direct fused pair; ragged lengths, NaN beyond every utterance, all epilogue modes."""
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


@pytest.fixture(params=[1, 2], ids=["two-6-wave-workgroups-per-CU", "one-12-wave-workgroup-per-CU"])
def chv(request, lib, experimental):  # (respair_wino_kernel failed its gate: DISSC_EXPERIMENTAL=1 builds only)
    """both workgroup shapes of the kernel (option "pairw_chv"); "pair_f23" off, so that mode 3 builds THIS kernel's form for
    C = 32, k = 11 too (the default there is the register-only F(2,3) pair, tested below)"""
    assert lib.lib.dissc_set_option(b"pairw_chv", request.param) == 0
    assert lib.lib.dissc_set_option(b"pair_f23", 0) == 0
    yield request.param
    lib.lib.dissc_set_option(b"pairw_chv", 2)
    lib.lib.dissc_set_option(b"pair_f23", F23_DEFAULT)


@pytest.mark.parametrize("C,k,d", SHAPES)
def test_fused_transform_domain_pair_matches_float64_and_the_other_paths(lib, chv, C, k, d):
    lengths = [2000, 1, 7, 255, 468, 469, 1023, 1999, 500, 12]
    ld = 2000
    x, w1, b1, w2, b2 = _data(C, k, lengths, ld, seed=C * 100 + k * 10 + d)
    ref = _reference(x, w1, b1, w2, b2, lengths, k, d)
    y3 = _pair(lib, 3, x, w1, b1, w2, b2, lengths, k, d)
    y0 = _pair(lib, 0, x, w1, b1, w2, b2, lengths, k, d)  # two direct launches
    worst3 = worst0 = 0.0
    for i, n in enumerate(lengths):
        assert torch.isfinite(y3[i, :, :n]).all()
        assert (y3[i, :, n:] == -7.0).all(), f"utterance {i}: wrote beyond its {n} samples"
        e3 = (y3[i, :, :n].double() - ref[i, :, :n]).abs().max().item()
        e0 = (y0[i, :, :n].double() - ref[i, :, :n]).abs().max().item()
        worst3, worst0 = max(worst3, e3), max(worst0, e0)
    r3 = float(((y3[0, :, :2000].double() - ref[0]) ** 2).mean().sqrt())
    r0 = float(((y0[0, :, :2000].double() - ref[0]) ** 2).mean().sqrt())
    print(f"C={C} k={k} d={d}: fused transform-domain pair max err {worst3:.2e} rms {r3:.2e}; two direct launches {worst0:.2e} / {r0:.2e}")
    assert worst3 <= 2e-5 and r3 <= max(3.0 * r0, 1e-6)
    # batch independence: an utterance alone gives the same bits
    for i in (3, 6):
        one = _pair(lib, 3, x[i:i + 1].clone(), w1, b1, w2, b2, lengths[i:i + 1], k, d)
        assert torch.equal(one[0, :, :lengths[i]], y3[i, :, :lengths[i]])
    # the two-launch transform-domain path (the product path of the 64-channel stage) does the same arithmetic in the
    # same order -- except that a tile's last F(4,3) groups see zeros where the two-launch path sees the neighbouring
    # tile's t (contributions that cancel exactly only in exact arithmetic): equal to rounding, not bit for bit
    if C == 64:
        y2 = _pair(lib, 2, x, w1, b1, w2, b2, lengths, k, d)
        for i, n in enumerate(lengths):
            assert (y2[i, :, :n] - y3[i, :, :n]).abs().max().item() <= 2e-6 if n else True


@pytest.mark.parametrize("C,k,d", [(32, 11, 3), (64, 3, 1), (32, 7, 5)])
def test_fused_transform_domain_pair_epilogue_modes(lib, chv, C, k, d):
    """MRF modes (acc = y | acc += y | acc = (acc + y) / 3) against the residual mode's y"""
    lengths = [700, 300, 1]
    x, w1, b1, w2, b2 = _data(C, k, lengths, 700, seed=7)
    y = _pair(lib, 3, x, w1, b1, w2, b2, lengths, k, d)
    acc0 = torch.rand(3, C, 700, device=DEV)
    for epi in (2, 3, 4):
        a = _pair(lib, 3, x, w1, b1, w2, b2, lengths, k, d, epi=epi, acc=acc0)
        for i, n in enumerate(lengths):
            want = y[i, :, :n] if epi == 2 else acc0[i, :, :n] + y[i, :, :n]
            if epi == 4:
                # on the CPU like the reference's xs / num_kernels: a true division (torch's GPU kernel for a scalar
                # divisor multiplies by the reciprocal; the HIP kernels use __fdiv_rn)
                want = (want.cpu() / 3.0).to(DEV)
            assert torch.equal(a[i, :, :n], want), (epi, i)
            assert torch.equal(a[i, :, n:], acc0[i, :, n:])


F23_DEFAULT = 3  # the "pair_f23" mask the library ships with (bit 0: C = 32, bit 1: C = 16)


@pytest.fixture
def f23(lib):
    """mode 3 of dissc_respair1d builds the register-only F(2,3) forms (respair_f23.hip, respair16_f23.hip) for k = 11"""
    assert lib.lib.dissc_set_option(b"pair_f23", 15) == 0  # (every shape with an instance)
    yield
    lib.lib.dissc_set_option(b"pair_f23", F23_DEFAULT)


@pytest.mark.parametrize("k", [11, 3])
@pytest.mark.parametrize("C,d", [(32, 1), (32, 3), (32, 5), (16, 1), (16, 3), (16, 5)])
def test_register_only_f23_pair_matches_float64_and_the_direct_pair(lib, f23, C, d, k):
    if k == 3:
        from conftest import is_experimental_build
        if not is_experimental_build():
            pytest.skip("the k = 3 instances measured neutral in the forward: DISSC_EXPERIMENTAL=1 builds only")
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
