"""Edge cases the reference's usage implies: empty / one-frame utterances in a batch, error
paths of the C ABI (never a crash, always a message), long inputs."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import dissc_amd
    from dissc_amd import _lib
    from oracle import generator_ref as gr
    import synthdata as synth
    sd = synth.synth_generator_state_dict(seed=0)
    g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to("cuda:0")
    g.load_state_dict(sd)
    g.eval().remove_weight_norm()
    return dict(g=g, gr=gr, synth=synth, folded=gr.fold_state_dict(sd), lib=_lib.lib, _lib=_lib, mod=dissc_amd)


def test_batch_with_empty_and_single_frame_utterances(env):
    g, gr, synth = env["g"], env["gr"], env["synth"]
    code, f0, spkr, _ = synth.synth_generator_inputs(4, 12, seed=42)
    lengths = np.array([12, 0, 1, 5], dtype=np.int32)
    y = g(code=torch.from_numpy(code), f0=torch.from_numpy(f0), spkr=torch.from_numpy(spkr),
          lengths=torch.from_numpy(lengths)).cpu()
    ref = gr.code_generator(env["folded"], synth.VCTK_CONFIG, code, f0, spkr, lengths=lengths)
    assert torch.isfinite(y).all()
    assert (y - ref).pow(2).mean().sqrt() <= 1e-4
    assert not y[1].any()


def test_dma_handover_of_wide_residual_pairs_on_a_poisoned_workspace(env):
    """Direct path (a generator built with option "wino" = 0), option "pair_dma" (default on): the first conv of a wide
    residual pair stores lrelu(t) with zero tails and the second stages its windows by LDS-DMA without masks, relying on
    those zeros (a row's tail is the next row's left halo).  With the scratch memory filled with NaN beforehand, a ragged
    batch must still give the bits of the masked path."""
    import dissc_amd
    synth, lib = env["synth"], env["lib"]
    B, T = 5, 70
    code, f0, spkr, _ = synth.synth_generator_inputs(B, T, seed=11)
    kw = dict(code=torch.from_numpy(code).cuda(), f0=torch.from_numpy(f0).cuda(), spkr=torch.from_numpy(spkr).cuda())
    lens = torch.tensor([70, 64, 33, 1, 17], dtype=torch.int32).cuda()
    outs = {}
    for v in (0, 1):  # a handle snapshots the options when it is created: one generator per setting
        try:
            assert lib.dissc_set_option(b"wino", 0) == 0 and lib.dissc_set_option(b"pair_dma", v) == 0
            g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to("cuda:0")
            g.load_state_dict(synth.synth_generator_state_dict(seed=0))
            g.eval().remove_weight_norm()
            g(**kw, lengths=lens)  # builds the native handle (every conv direct) and sizes the workspace
        finally:
            lib.dissc_set_option(b"wino", 1)
            lib.dissc_set_option(b"pair_dma", 1)
        g._ws.view(torch.float32)[: g._ws.numel() // 4].fill_(float("nan"))
        outs[v] = g(**kw, lengths=lens).clone()
        assert torch.isfinite(outs[v]).all(), v
    assert torch.equal(outs[0], outs[1])
    hop = outs[0].shape[-1] // T
    for b, n in enumerate([70, 64, 33, 1, 17]):
        assert not outs[1][b, 0, n * hop:].any()
    # the default generator (transform-domain convs) on a poisoned workspace: finite, nothing beyond the utterances
    gw = env["g"]
    gw(**kw, lengths=lens)
    gw._ws.view(torch.float32)[: gw._ws.numel() // 4].fill_(float("nan"))
    yw = gw(**kw, lengths=lens)
    assert torch.isfinite(yw).all()
    for b, n in enumerate([70, 64, 33, 1, 17]):
        assert not yw[b, 0, n * hop:].any()
    assert float((yw - outs[1]).abs().max()) <= 1e-4


def test_long_utterance_30s(env):
    """1500 frames (30 s): beyond the bench shape; finite, bounded, and batch independent."""
    g, synth = env["g"], env["synth"]
    code, f0, spkr, _ = synth.synth_generator_inputs(2, 1500, seed=9)
    lengths = torch.tensor([1500, 733], dtype=torch.int32)
    y = g(code=torch.from_numpy(code), f0=torch.from_numpy(f0), spkr=torch.from_numpy(spkr), lengths=lengths).cpu()
    assert y.shape == (2, 1, 480000) and torch.isfinite(y).all() and y.abs().max() <= 1.0
    y1 = g(code=torch.from_numpy(code[1:2, :733]), f0=torch.from_numpy(f0[1:2, :, :733]),
           spkr=torch.from_numpy(spkr[1:2])).cpu()
    assert torch.equal(y1[0, 0], y[1, 0, :733 * 320])


def test_out_of_range_ids_are_clamped_not_read_out_of_bounds(env):
    g, synth = env["g"], env["synth"]
    code, f0, spkr, _ = synth.synth_generator_inputs(1, 9, seed=3)
    code[0, 3] = 10 ** 9
    code[0, 4] = -5
    spkr[0, 0] = 9999
    # ids handed over from the host (files) raise like nn.Embedding does in the reference ...
    with pytest.raises(IndexError):
        g(code=torch.from_numpy(code), f0=torch.from_numpy(f0), spkr=torch.from_numpy(spkr))
    good = code.copy()
    good[0, 3] = good[0, 4] = 0
    with pytest.raises(IndexError):
        g(code=torch.from_numpy(good), f0=torch.from_numpy(f0), spkr=torch.from_numpy(spkr))  # speaker 9999
    # ... device-resident ids (not inspected: no sync) are clamped by the kernels, never read out of bounds
    y = g(code=torch.from_numpy(code).cuda(), f0=torch.from_numpy(f0), spkr=torch.from_numpy(spkr).cuda()).cpu()
    assert torch.isfinite(y).all()


def test_predictor_ids_and_f0_length_are_validated(env):
    import synthdata as synth
    from dissc_amd import predictors as P
    lm = P.LenPredictor(100, 108).to("cuda:0")
    lm.load_state_dict(synth.synth_len_state_dict(100, 108))
    with pytest.raises(IndexError):
        lm(torch.tensor([[1, 2, 500]]), torch.tensor([[3]]))
    with pytest.raises(IndexError):
        lm(torch.tensor([[1, 2, 3]]), torch.tensor([[108]]))
    pm = P.PitchPredictorBase(100, 108, id2pitch_mean=torch.zeros(10), id2pitch_std=torch.ones(10)).to("cuda:0")
    pm.load_state_dict(synth.synth_pitch_state_dict("base", 100, 108))
    with pytest.raises(IndexError):  # speaker 50 exists in the embedding but not in the 10-entry statistics
        pm.infer_freq(torch.tensor([[1, 2, 3]]), torch.tensor([[50]]), norm=False)
    y = pm.infer_freq(torch.tensor([[1, 2, 3]]).cuda(), torch.tensor([[50]]).cuda(), norm=False)  # clamped on device
    assert torch.isfinite(y).all()
    # non-divisible code / f0 lengths: the reference fails in torch.cat; we must not read OOB
    g = env["g"]
    code, f0, spkr, _ = env["synth"].synth_generator_inputs(2, 101, seed=3)
    with pytest.raises((RuntimeError, NotImplementedError)):
        g(code=torch.from_numpy(code), f0=torch.from_numpy(f0[:, :, :50]), spkr=torch.from_numpy(spkr))


def test_abi_error_paths_return_codes_and_messages(env):
    lib, _lib = env["lib"], env["_lib"]
    # workspace too small -> DISSC_ENOMEM, with a message, no launch
    g = env["g"]
    g._ensure()
    code = torch.zeros(1, 8, dtype=torch.int64, device="cuda")
    f0 = torch.zeros(1, 8, device="cuda")
    spk = torch.zeros(1, dtype=torch.int64, device="cuda")
    out = torch.zeros(1, 1, 8 * 320, device="cuda")
    ws = torch.empty(1024, dtype=torch.uint8, device="cuda")
    rc = lib.dissc_gen_forward(g._handle, code.data_ptr(), f0.data_ptr(), spk.data_ptr(), None, 1, 8,
                               out.data_ptr(), ws.data_ptr(), 1024, None)
    assert rc == -2 and b"workspace" in lib.dissc_last_error()
    rc = lib.dissc_gen_forward(g._handle, None, f0.data_ptr(), spk.data_ptr(), None, 1, 8, out.data_ptr(),
                               ws.data_ptr(), 1024, None)
    assert rc == -1
    # missing tensor at create -> DISSC_ENOTFOUND
    cfg = g._config()
    table, keep = _lib.make_tensor_table({"conv_pre.weight": torch.zeros(512, 257, 7)})
    h = ctypes.c_void_p()
    rc = lib.dissc_gen_create(ctypes.byref(cfg), table, 1, ctypes.byref(h))
    assert rc == -4 and b"missing" in lib.dissc_last_error()
    # unsupported conv geometry through the stand-alone entry point
    x = torch.zeros(1, 16, 64, device="cuda")
    w = torch.zeros(16, 16, 4)
    rc = lib.dissc_conv1d(x.data_ptr(), w.data_ptr(), None, x.data_ptr(), None, 1, 16, 16, 4, 1, 64, 64, 64,
                          ctypes.c_float(1.0), None)
    assert rc == -1
    with pytest.raises(env["mod"].DisscError):
        _lib.check(rc, "conv1d")


def test_predictor_and_hubert_limits(env):
    from dissc_amd import predictors as P
    from dissc_amd.hubert import HubertEncoder
    synth = env["synth"]
    pm = P.PitchPredictor(100, 108).to("cuda:0")
    pm.load_state_dict(synth.synth_pitch_state_dict("new", 100, 108))
    with pytest.raises(env["mod"].DisscError):  # > 850 frames: the reference's PE buffer ends too
        pm.infer_freq(torch.zeros(1, 851, dtype=torch.int64), torch.zeros(1, 1, dtype=torch.int64), True)
    pb = P.PitchPredictorBase(100, 108).to("cuda:0")
    pb.load_state_dict(synth.synth_pitch_state_dict("base", 100, 108))
    out = pb.infer_freq(torch.zeros(1, 1200, dtype=torch.int64), torch.zeros(1, 1, dtype=torch.int64), True)
    assert out.shape == (1, 1200) and torch.isfinite(out).all()
    enc = HubertEncoder(synth.synth_hubert_state_dict(6), synth.synth_kmeans_centers(), 6).to("cuda:0")
    with pytest.raises(ValueError):
        enc(torch.zeros(1, 399))
    assert P.infer_samples([], [], None, None) == []


def test_maximum_sizes_index_arithmetic():
    """Activation buffers beyond 2^31 ELEMENTS (generator: 1 700 x 5 s, 2.2e9 floats per stage buffer; HuBERT: 170 x 10 s,
    2.8e9 floats out of conv0): rows of the giant batch are bit-identical to the same utterance on its own, tails are zero --
    an int32 offset anywhere in the kernels or the host planners would show here (tools/max_size_probe.py has the larger sweep)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if torch.cuda.get_device_properties(0).total_memory < 150 * 2 ** 30:
        pytest.skip("needs ~100 GB of HBM")
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "max_size_probe", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "max_size_probe.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    g = m.make_generator()
    assert m.gen_case(g, 1700, 250, [0, 3, 849, 1342, 1343, 1699], True)
    del g
    torch.cuda.empty_cache()
    enc = m.make_encoder()
    assert m.hubert_case(enc, 170, 10.0, rows=(0, 85, 169))
    del enc
    torch.cuda.empty_cache()
