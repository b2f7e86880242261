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
from test_gpu_generator import env, _generator_with, _pair_cases  # noqa: E402,F401


def test_transform_domain_pairs_agree_with_the_unfused_generator(env, experimental):
    """respair_wino.hip (opt-in, "pair_wino" = 1: the k = 11, d = 1 / 3 pairs of the 32-channel stage and the first k = 3
    pair of the 64-channel stage as ONE transform-domain launch each; = 2: every shape with an instance) against the
    default instance: same waveform to fp32 rounding, ragged and at the BASELINE size; fewer executed FLOPs are
    reported for them."""
    lib, synth = env["lib"], env["synth"]
    # (without the forms that took those shapes over since: the register-only F(2,3) pairs and the k = 3 layers on conv_wino8)
    base = dict(pair_f23=0, wino8_mask=0o770770770)
    gd = _generator_with(lib, synth, pair_wino=0, **base)
    g1 = _generator_with(lib, synth, pair_wino=1, **base)   # the shapes that measured faster per launch
    ga = _generator_with(lib, synth, pair_wino=2, **base)   # every shape that has an instance
    assert g1.flops_executed(1000) < gd.flops_executed(1000) and g1.flops(1000) == gd.flops(1000)
    assert ga.flops_executed(1000) < g1.flops_executed(1000)
    for g, (code, f0, spkr, lengths) in [(g_, c) for g_ in (g1, ga) for c in _pair_cases(synth)]:
        kw = dict(code=torch.from_numpy(code), f0=torch.from_numpy(f0), spkr=torch.from_numpy(spkr),
                  lengths=torch.from_numpy(lengths))
        yw, yd = g(**kw).cpu(), gd(**kw).cpu()
        assert torch.isfinite(yw).all() and not torch.equal(yw, yd)
        e = (yw - yd).double()
        rms = float(e.pow(2).mean().sqrt())
        print(f"B={code.shape[0]} T={code.shape[1]}: fused transform-domain pairs vs pair_wino=0: rms {rms:.2e}, max {float(e.abs().max()):.2e}")
        assert rms <= 5e-6 and float(e.abs().max()) <= 1e-4
        one = g(code=kw["code"][:1], f0=kw["f0"][:1], spkr=kw["spkr"][:1], lengths=kw["lengths"][:1]).cpu()[0]
        assert torch.equal(one, yw[0])  # an utterance's samples do not depend on the batch it runs in
