"""Round 6 kernels of the HuBERT encoder through the C ABI (csrc/lin_gemm.hip):
  lin128_kernel     -- the linears (1x1 convs) on 128 / 256 x 128 tiles: BIT-IDENTICAL to conv_mfma32_kernel's 256 x 64 instances in every
                       tile mode (same MFMA sequence over k per output, same epilogue arithmetic), ragged batches, NaN-poisoned padding;
  conv2s128_kernel  -- the stride-2, k = 3 feature convs: another summation order, so held to a float64 convolution at the error level of
                       the kernel it replaces, not to its bits.
Reference: fairseq's TransformerSentenceEncoderLayer linears / ConvFeatureExtractionModel as called behind data/encode.py:21-22,32."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from dissc_amd import _lib
    return _lib


def _lin(L, x, w, bias, lengths, mode):
    assert L.lib.dissc_set_option(b"lin128", mode) == 0
    try:
        B, cin, T = x.shape
        cout = w.shape[0]
        ld = (T + 3) // 4 * 4
        xd = torch.full((B, cin, ld), float("nan"), device="cuda")
        for i, n in enumerate(lengths):
            xd[i, :, :n] = x[i, :, :n].cuda()
        yd = torch.full((B, cout, ld), -7.0, device="cuda")
        ln = torch.as_tensor(lengths, dtype=torch.int32).cuda()
        wc, bc = w.contiguous(), bias.contiguous()
        L.check(L.lib.dissc_conv1d(xd.data_ptr(), wc.data_ptr(), bc.data_ptr(), yd.data_ptr(), ln.data_ptr(), B, cin, cout, 1, 1, ld, ld,
                                   T, ctypes.c_float(1.0), None), "dissc_conv1d")
        torch.cuda.synchronize()
        return yd.cpu()
    finally:
        L.lib.dissc_set_option(b"lin128", 1)


@pytest.mark.parametrize("cin,cout,T,lengths", [
    (768, 3072, 499, [499, 0, 1, 3, 127, 128, 129, 257, 498]),   # fc1's shape: tile edges, an empty and a one-frame utterance
    (3072, 768, 300, [300, 128, 129, 5]),                         # fc2's: the long K loop
    (512, 768, 131, [131, 4, 2]),                                 # post_extract_proj's
    (768, 2304, 260, [260, 256, 255]),                            # qkv's
])
def test_lin128_is_bit_identical_to_the_256x64_kernel_in_every_tile_mode(L, cin, cout, T, lengths):
    rs = np.random.RandomState(cin + cout + T)
    x = torch.from_numpy(rs.standard_normal((len(lengths), cin, T)).astype(np.float32))
    w = torch.from_numpy((rs.standard_normal((cout, cin, 1)) / np.sqrt(cin)).astype(np.float32))
    bias = torch.from_numpy(rs.standard_normal(cout).astype(np.float32))
    y0 = _lin(L, x, w, bias, lengths, 0)
    ref = torch.einsum("oc,bcl->bol", w[:, :, 0].double(), x.double()) + bias.double()[None, :, None]
    for i, n in enumerate(lengths):
        if n:
            assert float((y0[i, :, :n].double() - ref[i, :, :n]).abs().max()) <= 3e-5
    for mode in (1, 2, 3, 5):  # the per-launch policy and the three forced tile shapes
        y = _lin(L, x, w, bias, lengths, mode)
        for i, n in enumerate(lengths):
            assert torch.equal(y[i, :, :n], y0[i, :, :n]), (mode, i, n)
            assert bool((y[i, :, n:] == -7.0).all()), f"mode {mode}: utterance {i} wrote beyond its {n} frames"


def _conv_s2(L, x, w, bias, lengths, mode, act):
    assert L.lib.dissc_set_option(b"conv2s128", mode) == 0
    try:
        B, cin, T = x.shape
        cout = w.shape[0]
        To = (T - 3) // 2 + 1
        ldx, ldo = (T + 3) // 4 * 4, (To + 3) // 4 * 4
        xd = torch.full((B, cin, ldx), float("nan"), device="cuda")
        for i, n in enumerate(lengths):
            xd[i, :, :n] = x[i, :, :n].cuda()
        yd = torch.full((B, cout, ldo), -7.0, device="cuda")
        ln = torch.as_tensor(lengths, dtype=torch.int32).cuda()
        wc, bc = w.contiguous(), bias.contiguous()
        L.check(L.lib.dissc_conv1d_s2(xd.data_ptr(), wc.data_ptr(), bc.data_ptr(), yd.data_ptr(), ln.data_ptr(), B, cin, cout, 3, ldx, ldo,
                                      T, act, 0, None), "dissc_conv1d_s2")
        torch.cuda.synchronize()
        return yd.cpu()
    finally:
        L.lib.dissc_set_option(b"conv2s128", 1)


@pytest.mark.parametrize("cin,cout,T,lengths,act", [
    (512, 512, 1031, [1031, 3, 4, 5, 259, 516, 777, 1030], 1),   # HuBERT's shape: one output, tile edges (257 / 258 outputs), odd and even tails
    (64, 256, 300, [300, 257, 2], 0),                             # a short K loop; an utterance too short for one output
    (512, 256, 4099, [4099], 1),                                  # more than one sweep of column tiles, one M tile
])
def test_conv2s128_matches_float64_like_the_kernel_it_replaces(L, cin, cout, T, lengths, act):
    rs = np.random.RandomState(cin + cout + T)
    x = torch.from_numpy(rs.standard_normal((len(lengths), cin, T)).astype(np.float32))
    w = torch.from_numpy((rs.standard_normal((cout, cin, 3)) / np.sqrt(3 * cin)).astype(np.float32))
    bias = torch.from_numpy(rs.standard_normal(cout).astype(np.float32))
    ys = {m: _conv_s2(L, x, w, bias, lengths, m, act) for m in (0, 1, 3)}
    for i, n in enumerate(lengths):
        no = (n - 3) // 2 + 1 if n >= 3 else 0
        for m, y in ys.items():
            assert bool((y[i, :, no:] == -7.0).all()), f"conv2s128={m}: utterance {i} wrote beyond its {no} outputs"
        if not no:
            continue
        ref = F.conv1d(x[i:i + 1, :, :n].double(), w.double(), bias.double(), stride=2)
        ref = F.gelu(ref)[0] if act else ref[0]
        err = {m: float((y[i, :, :no].double() - ref).abs().max()) for m, y in ys.items()}
        rms = {m: float((y[i, :, :no].double() - ref).pow(2).mean().sqrt()) for m, y in ys.items()}
        assert max(err.values()) <= 2e-5, (i, err)
        assert rms[1] <= 1.5 * rms[0] + 1e-9 and rms[3] <= 1.5 * rms[0] + 1e-9, (i, rms)  # same error level as the 256 x 64 kernel
        assert torch.equal(ys[1][i, :, :no], ys[3][i, :, :no])  # 16 or 32 channels per barrier: the same summation order, the same bits


def test_encoder_units_and_features_do_not_depend_on_the_linear_kernel(L):
    """HubertEncoder handles created under lin128 = 0 and = 1 (options are frozen per handle): bitwise the same dense features and units --
    the linears are a schedule, not an arithmetic; ragged batch with NaN padding."""
    from dissc_amd.hubert import HubertEncoder
    import synthdata as synth
    sd, centers = synth.synth_hubert_state_dict(6), synth.synth_kmeans_centers()
    ns = [48000, 16000, 4000, 719]
    wav = torch.full((len(ns), max(ns)), float("nan"))
    for i, n in enumerate(ns):
        wav[i, :n] = torch.from_numpy(synth.synth_waveform(n, seed=40 + i))
    outs = []
    for mode in (0, 1):
        assert L.lib.dissc_set_option(b"lin128", mode) == 0
        try:
            enc = HubertEncoder(sd, centers, n_layers=6).to("cuda:0")
            outs.append(enc(wav, n_samples=torch.tensor(ns)))
        finally:
            L.lib.dissc_set_option(b"lin128", 1)
    for i in range(len(ns)):
        T = int(outs[0]["frames"][i])
        assert torch.equal(outs[0]["units"][i, :T], outs[1]["units"][i, :T])
        assert torch.equal(outs[0]["dense"][i, :T], outs[1]["dense"][i, :T])
