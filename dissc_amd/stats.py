"""Per-speaker F0 statistics on the GPU: the reduction behind
``data/data_utils.calculate_pitch_stats`` (reference data/data_utils.py:33-46).

The JSONL parse stays on the host; the frames are grouped by speaker into one fp64 buffer and the
HIP kernel (csrc/pitch_stats.hip, one workgroup per speaker, fixed reduction tree) returns mean
and population std of the voiced (non-zero) frames.
"""
import numpy as np
import torch

from ._lib import check, current_stream_ptr, lib


class NoVoicedFramesError(ValueError):
    """A speaker has no voiced (non-zero) F0 frame: its mean/std would be 0/0."""


def pitch_stats(f0_by_speaker, device="cuda:0", on_unvoiced="nan"):
    """f0_by_speaker: {speaker: sequence of floats (0 = unvoiced)} in insertion order
    -> {speaker: {'mean': np.float64, 'std': np.float64}} (same schema as the reference's pickle).
    on_unvoiced: what to do with a speaker without a single voiced frame -- "nan" returns NaN
    statistics like the reference's numpy (with its RuntimeWarning), "raise" raises
    NoVoicedFramesError naming the speakers (the kernel returns the voiced count)."""
    names = list(f0_by_speaker.keys())
    if not names:
        return {}
    arrays = [np.asarray(f0_by_speaker[k], dtype=np.float64).reshape(-1) for k in names]
    offsets = np.zeros(len(names) + 1, dtype=np.int64)
    np.cumsum([a.size for a in arrays], out=offsets[1:])
    flat = np.concatenate(arrays) if offsets[-1] else np.zeros(0, dtype=np.float64)
    dev = torch.device(device)
    with torch.cuda.device(dev):
        d_f0 = torch.from_numpy(flat).to(dev) if flat.size else torch.zeros(1, dtype=torch.float64, device=dev)
        d_off = torch.from_numpy(offsets).to(dev)
        mean = torch.empty(len(names), dtype=torch.float64, device=dev)
        std = torch.empty_like(mean)
        cnt = torch.empty(len(names), dtype=torch.int64, device=dev)
        check(lib.dissc_pitch_stats(d_f0.data_ptr(), d_off.data_ptr(), len(names), mean.data_ptr(),
                                    std.data_ptr(), cnt.data_ptr(), current_stream_ptr()), "dissc_pitch_stats")
        mean, std, cnt = mean.cpu().numpy(), std.cpu().numpy(), cnt.cpu().numpy()
    if on_unvoiced == "raise":
        bad = [k for i, k in enumerate(names) if cnt[i] == 0]
        if bad:
            raise NoVoicedFramesError(
                f"{len(bad)} of {len(names)} speakers have no voiced F0 frame (e.g. {bad[:5]}): the F0 track of "
                "this file is empty/all-zero (written by data/encode.py --f0 zeros?); statistics would be NaN")
    elif on_unvoiced != "nan":
        raise ValueError("on_unvoiced must be 'nan' or 'raise'")
    return {k: {"mean": mean[i], "std": std[i]} for i, k in enumerate(names)}
