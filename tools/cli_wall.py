#!/usr/bin/env python
"""Wall time of the CLI a user actually runs, from process start to the last file closed.

    python tools/cli_wall.py [--utts 2592] [--targets 4] [--ranks 1] [--out profiles/r04/cli_wall.json]

Builds a configs[4]-shaped manifest (default 108 speakers x 24 utterances of 2-5 s, x 4 target speakers = 10 368
generator jobs, ~36 000 s of audio, ~2.3 GB of float32 WAVs), then runs `sr/inference.py` on it as a fresh process
(`--ranks N` > 1: N gloo ranks sharing this GPU -- a functional rehearsal of the N-writer path, not a speed claim) with
DISSC_CLI_TIMING=1, which makes the CLI print its own phase split measured from the process creation time:
imports, process group, HIP init, manifest + checkpoint load, weight fold + pack, job list, resynthesis + writes,
ground-truth copies, teardown.  The outer wall (fork -> exit) is taken here."""
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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=2592)
    ap.add_argument("--targets", type=int, default=4)
    ap.add_argument("--ranks", type=int, nargs="+", default=[1])
    ap.add_argument("--writers", default="all")
    ap.add_argument("--out", default=None)
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    import synthdata as synth
    td = tempfile.mkdtemp(prefix="dissc_cli_wall_")
    os.makedirs(f"{td}/ckpt")
    os.makedirs(f"{td}/meta")
    shutil.copy(os.path.join(ROOT, "tests", "golden", "vctk_id_to_spkr.pkl"), f"{td}/meta/id_to_spkr.pkl")
    import pickle
    ids = pickle.load(open(f"{td}/meta/id_to_spkr.pkl", "rb"))
    cfg = dict(synth.VCTK_CONFIG, input_training_file=f"{td}/meta/train.txt", f0_normalize=False, f0_stats=None)
    json.dump(cfg, open(f"{td}/ckpt/config.json", "w"))
    torch.save({"generator": synth.synth_generator_state_dict(seed=0)}, f"{td}/ckpt/g_00000001")
    rs = np.random.RandomState(3)
    frames = 0
    with open(f"{td}/man.txt", "w") as f:
        for u in range(a.utts):
            T = int(rs.randint(100, 251))
            code, f0, _, _ = synth.synth_generator_inputs(1, T, seed=5000 + u)
            frames += T
            f.write(json.dumps({"units": code[0].tolist(), "f0": [float(v) for v in f0[0, 0]],
                                "audio": f"{ids[u % len(ids)]}_{u:05d}.wav"}) + "\n")
    targets = ids[3:3 + a.targets]
    audio_sec = frames * a.targets * 0.02
    base = ["--input_code_file", f"{td}/man.txt", "--data_path", f"{td}/nowav", "--checkpoint_file", f"{td}/ckpt/", "--vc",
            "--target-speakers"] + targets + ["--unseen_speaker", "--id_to_spkr", f"{td}/meta/id_to_spkr.pkl", "-n", "-1"]
    results = []
    for n in a.ranks:
        out_dir = f"{td}/out{n}"
        env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")},
                   DISSC_CLI_TIMING="1", DISSC_WRITERS=a.writers)
        cli = os.path.join(ROOT, "sr", "inference.py")
        if n == 1:
            cmd = [sys.executable, cli]
        else:
            env["DISSC_DIST_BACKEND"] = "gloo"
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
                   "127.0.0.1", "--master-port", str(29700 + n), cli]
        for rep in range(2):  # second run: page cache warm, like a user's second invocation
            shutil.rmtree(out_dir, ignore_errors=True)
            t0 = time.time()
            r = subprocess.run(cmd + base + ["--output_dir", out_dir], env=env, capture_output=True, text=True, cwd=td)
            wall = time.time() - t0
            if r.returncode != 0:
                raise SystemExit(r.stdout[-2000:] + r.stderr[-4000:])
            # (the ranks of a multi-process run share the pipe: two records can land on one line)
            phases, dec, pos = [], json.JSONDecoder(), 0
            while True:
                pos = r.stdout.find("CLI_TIMING {", pos)
                if pos < 0:
                    break
                obj, end = dec.raw_decode(r.stdout, pos + len("CLI_TIMING "))
                phases.append(obj)
                pos = end
            files = [f for f in os.listdir(out_dir) if f.endswith("_gen.wav")]
            nbytes = sum(os.path.getsize(os.path.join(out_dir, f)) for f in files)
            rec = {"ranks": n, "rep": rep, "writers": a.writers if n > 1 else "rank0 (N = 1)", "outer_wall_s": round(wall, 3),
                   "files": len(files), "bytes_written": nbytes, "audio_sec": round(audio_sec, 1),
                   "x_real_time_end_to_end": round(audio_sec / wall, 1), "phases": sorted(phases, key=lambda p: p["rank"])}
            print(json.dumps(rec), flush=True)
            results.append(rec)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump({"what": __doc__.split("\n\n")[0], "utts": a.utts, "targets": a.targets, "runs": results},
                  open(a.out, "w"), indent=1)
    if not a.keep:
        shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    main()
