"""Per-speaker F0 statistics on the GPU: the reduction behind
``data/data_utils.calculate_pitch_stats`` (reference data/data_utils.py:33-46).

The JSONL parse stays on the host; the frames are grouped by speaker into one fp64 buffer and the
HIP kernel (csrc/pitch_stats.hip, one workgroup per speaker, fixed reduction tree) returns mean
and population std of the voiced (non-zero) frames.
"""
import numpy as np
import torch

from ._lib import check, current_stream_ptr, lib


def pitch_stats(f0_by_speaker, device="cuda:0"):
    """f0_by_speaker: {speaker: sequence of floats (0 = unvoiced)} in insertion order
    -> {speaker: {'mean': np.float64, 'std': np.float64}} (same schema as the reference's pickle)."""
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
        mean, std = mean.cpu().numpy(), std.cpu().numpy()
    return {k: {"mean": mean[i], "std": std[i]} for i, k in enumerate(names)}
