#!/usr/bin/env python
"""Writes tests/golden/yaapt_speech.npz: the F0 tracks (5 ms hop, Hz, 0 = unvoiced) that THIS repo's CPU
restatement of YAAPT (oracle/yaapt_ref.py, parity with amfm_decompy UNPINNED) produces for the reference's two
speech fixtures s1_1.wav / s1_2.wav.  These are not reference outputs (amfm_decompy is absent offline): the file
exists so that drift of the restatement or of the HIP tracker on real speech is visible in review.

    python tests/golden/make_yaapt_tracks.py
"""
import os
import sys

import numpy as np
from scipy.io import wavfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import yaapt_ref as yr  # noqa: E402

if __name__ == "__main__":
    out = {}
    for name in ("s1_1", "s1_2"):
        sr, x = wavfile.read(os.path.join(HERE, name + ".wav"))
        assert sr == 16000 and x.dtype == np.int16
        out[name] = yr.get_yaapt_f0(x.astype(np.float32) / 32768.0).astype(np.float32)
        print(name, len(out[name]), "frames, voiced", float((out[name] > 0).mean()))
    np.savez_compressed(os.path.join(HERE, "yaapt_speech.npz"), **out)
