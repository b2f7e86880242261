"""Tests of the kernels that only DISSC_EXPERIMENTAL=1 builds carry (their gates failed; experimental/csrc, and the
-DDISSC_EXPERIMENTAL=1 parts of dissc_amd/csrc): moved out of tests/ in round 6 so that the default suite does not collect tests it
can only skip.  Run on a GPU box after `DISSC_EXPERIMENTAL=1 python -c "import __graft_entry__ as g; g.build()"`:
    python -m pytest experimental/tests -m gpu -q"""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
pytestmark = pytest.mark.gpu
from test_gpu_edge_cases import env  # noqa: E402,F401


def test_hipgraph_replay_is_bit_identical(env, experimental):
    """option "graphs" (off by default: measured slower on ROCm 7.2): a small forward captured into a hipGraph --
    the three ResBlock streams join the capture through their events -- and replayed gives the same bits, also when
    the ragged lengths behind the same pointer change between replays"""
    import dissc_amd
    g0, synth, lib = env["g"], env["synth"], env["lib"]
    code, f0, spkr, _ = synth.synth_generator_inputs(3, 60, seed=7)
    kw = dict(code=torch.from_numpy(code).cuda(), f0=torch.from_numpy(f0).cuda(), spkr=torch.from_numpy(spkr).cuda())
    lens = torch.tensor([60, 41, 13], dtype=torch.int32).cuda()
    plain = [g0(**kw, lengths=lens).clone()]
    lens.copy_(torch.tensor([22, 60, 5], dtype=torch.int32))
    plain.append(g0(**kw, lengths=lens).clone())
    hits0 = ctypes.c_int()
    lib.dissc_get_option(b"graph_hits", ctypes.byref(hits0))
    try:  # a handle created under "graphs" = 1
        assert lib.dissc_set_option(b"graphs", 1) == 0
        g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to("cuda:0")
        g.load_state_dict(synth.synth_generator_state_dict(seed=0))
        g.eval().remove_weight_norm()
        lens.copy_(torch.tensor([60, 41, 13], dtype=torch.int32))
        g(**kw, lengths=lens)
    finally:
        assert lib.dissc_set_option(b"graphs", 0) == 0
    for rep in range(4):
        y = g(**kw, lengths=lens)
        assert torch.equal(y, plain[0]), rep
        del y
    lens.copy_(torch.tensor([22, 60, 5], dtype=torch.int32))
    assert torch.equal(g(**kw, lengths=lens), plain[1])
    hits = ctypes.c_int()
    lib.dissc_get_option(b"graph_hits", ctypes.byref(hits))
    assert hits.value > hits0.value  # replays happened (the caching allocator hands the same buffers back)
