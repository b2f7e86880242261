"""ORACLE (test infrastructure only): CPU restatement of the reference's per-speaker F0 statistics.

Follows reference data/data_utils.py:33-46 (calculate_pitch_stats): frames of all utterances of a
speaker (speaker = text before the first '_' of the 'audio' field) are concatenated in file order,
unvoiced frames (f0 == 0) dropped, and numpy's fp64 ``mean()`` / ``std()`` (population, ddof = 0)
taken.  Pinned against the reference itself by tests/golden/prep_expected.pkl
(tests/golden/make_golden.py prep_dataset).  Only tests/ may import this module.
"""
import numpy as np


def pitch_stats(records):
    """records: iterable of dicts with 'audio' and 'f0' -> {speaker: {'mean', 'std'}}"""
    by_spk = {}
    for r in records:
        by_spk.setdefault(r["audio"].split("_")[0], []).extend(r["f0"])
    out = {}
    for k, v in by_spk.items():
        a = np.array(v, dtype=np.float64)
        a = a[a != 0]
        out[k] = {"mean": a.mean(), "std": a.std()}
    return out
