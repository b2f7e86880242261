#!/usr/bin/env python
"""Wall time of `data/encode.py` (wav directory -> units / F0 JSONL) on a synthetic corpus, process start to manifest
closed, with the CLI's own phase split (DISSC_CLI_TIMING=1).

    python tools/encode_wall.py [--utts 2592] [--out profiles/r04/encode_wall.json]

The corpus: `--utts` 16-bit wavs of 2-5 s (speech-like: a harmonic source with a moving F0, amplitude-modulated, plus
noise -- so that YAAPT tracks something), synthetic HuBERT / k-means checkpoints (synthdata.py)."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
from scipy.io import wavfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def speechlike(rs, n):
    t = np.arange(n) / 16000.0
    f0 = 120.0 + 60.0 * rs.rand() + 30.0 * np.sin(2 * np.pi * (0.5 + rs.rand()) * t)
    ph = 2 * np.pi * np.cumsum(f0) / 16000.0
    x = sum(np.sin(k * ph) / k for k in range(1, 8))
    env = 0.5 * (1 + np.sin(2 * np.pi * 2.5 * t + rs.rand() * 6.28)) ** 2
    x = 0.2 * x * env + 0.01 * rs.randn(n)
    return (np.clip(x, -1, 1) * 32767).astype(np.int16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=2592)
    ap.add_argument("--out", default=None)
    ap.add_argument("--f0", default="yaapt")
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    import synthdata as synth
    td = tempfile.mkdtemp(prefix="dissc_encode_wall_")
    os.makedirs(f"{td}/ckpt")
    os.makedirs(f"{td}/wav")
    torch.save({"model": synth.synth_hubert_state_dict(6)}, f"{td}/ckpt/hubert-base-ls960.pt")
    np.save(f"{td}/ckpt/kmeans_100.npy", synth.synth_kmeans_centers().numpy())
    rs = np.random.RandomState(11)
    total = 0
    for u in range(a.utts):
        n = int(rs.randint(32000, 80001))
        total += n
        wavfile.write(f"{td}/wav/u{u:05d}.wav", 16000, speechlike(rs, n))
    audio_sec = total / 16000.0
    runs = []
    for rep in range(a.reps):
        out_file = f"{td}/out{rep}/enc.txt"
        env = dict(os.environ, DISSC_CLI_TIMING="1")
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "data", "encode.py"), "--base_dir", f"{td}/wav", "--out_file",
                            out_file, "--checkpoint_dir", f"{td}/ckpt", "--f0", a.f0], capture_output=True, text=True, env=env)
        wall = time.time() - t0
        if r.returncode != 0:
            raise SystemExit(f"encode.py failed:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}")
        phases = [json.loads(l[len("CLI_TIMING "):]) for l in r.stdout.splitlines() if l.startswith("CLI_TIMING ")]
        lines = sum(1 for _ in open(out_file))
        assert lines == a.utts, (lines, a.utts)
        runs.append({"rep": rep, "outer_wall_s": round(wall, 3), "x_real_time_end_to_end": round(audio_sec / wall, 1),
                     "manifest_bytes": os.path.getsize(out_file), "phases": phases})
        print(json.dumps(runs[-1]), flush=True)
    rec = {"what": "data/encode.py end to end: wav directory -> units / F0 manifest, process start to manifest closed",
           "utts": a.utts, "audio_sec": round(audio_sec, 1), "f0": a.f0, "runs": runs}
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rec, f, indent=1)
    shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    main()
