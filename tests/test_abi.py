"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/dissc_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    import dissc_amd
    return dissc_amd


def _declared_symbols():
    syms = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            src = open(os.path.join(ROOT, "include", fn)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            syms.update(re.findall(r"\b(dissc_[a-z0-9_]+)\s*\(", src))
    return syms


def test_library_exports_every_declared_symbol(built):
    L = ctypes.CDLL(built.library_path())
    syms = _declared_symbols()
    assert len(syms) >= 10
    for s in sorted(syms):
        assert hasattr(L, s), f"{s} declared in include/*.h but not exported"


def test_abi_version_and_error_string(built):
    assert built.lib.dissc_abi_version() >= 1
    assert built.lib.dissc_last_error() is not None


def test_generator_wrapper_host_logic(built):
    import torch
    import synthdata as synth
    g = built.CodeGenerator(synth.VCTK_CONFIG)
    sd = synth.synth_generator_state_dict(0)
    g.load_state_dict(sd)
    g.eval().remove_weight_norm()
    assert len(g._folded) == 97 * 2 + 2
    # folded ConvTranspose weight keeps the [Cin, Cout, k] layout
    assert tuple(g._folded["ups.0.weight"].shape) == (512, 256, 11)
    bad = dict(sd)
    del bad["ups.3.weight_g"]
    with pytest.raises(RuntimeError):
        built.CodeGenerator(synth.VCTK_CONFIG).load_state_dict(bad)
    with pytest.raises(NotImplementedError):
        built.CodeGenerator(dict(synth.VCTK_CONFIG, lambda_commit=0.1))
    with pytest.raises(built.DisscError):
        g.to("cpu")  # no CPU fallback
    assert torch.equal(g._upsample(torch.arange(3.).view(1, 1, 3), 6),
                       torch.tensor([0., 0, 1, 1, 2, 2]).view(1, 1, 6))


def test_collective_entries_refuse_bad_arguments_without_a_gpu(built):
    """dissc_comm_* / dissc_allgather_waves (SURVEY 8(b)'s collective): argument errors are reported, nothing is launched."""
    import dissc_amd.collective as coll
    L = built.lib
    assert L.dissc_allgather_waves(None, None, 0, None, None) == -1 and b"bad argument" in L.dissc_last_error()
    assert L.dissc_comm_unique_id(None) == -1
    h = ctypes.c_void_p()
    assert L.dissc_comm_create(b"\0" * 128, 0, 0, ctypes.byref(h)) == -1 and h.value is None
    assert L.dissc_comm_create(b"\0" * 128, 2, 2, ctypes.byref(h)) == -1
    assert L.dissc_comm_destroy(None) == 0
    with pytest.raises(ValueError):
        coll.WaveComm(b"short", 1, 0, device="cuda:0")


def test_per_source_compiler_flags_name_existing_sources():
    """__graft_entry__.EXTRA_FLAGS keys are file names of dissc_amd/csrc: a renamed source must not silently lose its flag
    (attn.hip is only as fast as measured with -amdgpu-mfma-vgpr-form: without it the compiler moves the accumulators through AGPRs)."""
    import __graft_entry__ as ge
    csrc = os.path.join(ROOT, "dissc_amd", "csrc")
    assert "attn.hip" in ge.EXTRA_FLAGS and "-amdgpu-mfma-vgpr-form" in ge.EXTRA_FLAGS["attn.hip"]
    for name, flags in ge.EXTRA_FLAGS.items():
        assert os.path.exists(os.path.join(csrc, name)), name
        assert isinstance(flags, list) and all(isinstance(f, str) for f in flags)
