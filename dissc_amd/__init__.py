"""dissc_amd -- MI355X (gfx950) implementation of the DISSC inference hot path.

Python here is glue only: it mirrors the reference's module signatures
(CodeGenerator, LenPredictor, PitchPredictor, SpeechEncoder) and forwards to the
hand-written HIP kernels in libdissc_hip.so through the C ABI declared in
include/dissc_hip.h.  There is no CPU fallback: importing the compute classes
without the built library raises.
"""
from ._lib import lib, DisscError, library_path  # noqa: F401
from .generator import CodeGenerator, AttrDict  # noqa: F401

__all__ = ["lib", "DisscError", "library_path", "CodeGenerator", "AttrDict"]
