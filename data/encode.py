#!/usr/bin/env python
"""wav directory -> units JSONL with the HuBERT-base + k-means-100 encoder on MI355X.

Same command line and output format as the reference's data/encode.py (reference
data/encode.py:11-41): one line ``{"units": [...], "f0": [...], "durations": [...],
"audio": file}`` per wav in ``--base_dir`` (os.listdir order), appended to ``--out_file``.

Differences in execution: files are encoded in length-sorted batches through the HIP encoder
(per-utterance exact) instead of one B=1 call each; lines are appended batch by batch.
Checkpoints are not downloaded (no network): ``--checkpoint_dir`` / $DISSC_CHECKPOINT_DIR must
hold ``<model_name>.pt`` and ``<quantizer_name>_<vocab_size>.{npy,bin,pt}``.
``f0`` is tracked with YAAPT on the GPU (dissc_amd/f0.py; parity with amfm_decompy unpinned, see
oracle/yaapt_ref.py); ``--f0 zeros`` (explicit, with a warning) skips it and writes an all-unvoiced track, valid
only for the --pred_pitch flows where infer.py never reads it (reference infer.py:36-39,149-155).
"""
import argparse
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch
from scipy.io import wavfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_wav(path):
    """torchaudio.load semantics for 16-bit PCM: float32 in [-1, 1), [1, N] (first channel)."""
    sr, x = wavfile.read(path)
    if x.ndim > 1:
        x = x[:, 0]
    if x.dtype == np.int16:
        x = x.astype(np.float32) / 32768.0
    elif x.dtype == np.int32:
        x = x.astype(np.float32) / 2147483648.0
    elif x.dtype == np.uint8:
        x = (x.astype(np.float32) - 128.0) / 128.0
    else:
        x = x.astype(np.float32)
    return x, sr


def wav_frames(path):
    """sample count from the header (falls back to reading the file)"""
    import wave
    try:
        with wave.open(path, "rb") as w:
            return w.getnframes()
    except (wave.Error, EOFError):
        return len(load_wav(path)[0])


_TRACKER = {}


class _Timing:
    """DISSC_CLI_TIMING=1 (tools/encode_wall.py): seconds per phase, printed as one `CLI_TIMING {json}` line at exit"""

    def __init__(self):
        self.on = os.environ.get('DISSC_CLI_TIMING') == '1'
        self.acc = {}
        if self.on:
            import time

            import psutil
            self.now = time.time
            self.last = psutil.Process().create_time()
            self.add('imports')

    def add(self, name):
        if self.on:
            t = self.now()
            self.acc[name + '_s'] = self.acc.get(name + '_s', 0.0) + t - self.last
            self.last = t

    def dump(self, **extra):
        if self.on:
            out = {k: round(v, 4) for k, v in self.acc.items()}
            print('CLI_TIMING ' + json.dumps(dict(out, total_s=round(sum(self.acc.values()), 4), **extra)), flush=True)


def track_f0(wav, ns, frames, args):
    """F0 per unit frame for a batch: YAAPT at a 5 ms hop on the GPU (dissc_amd/f0.py), then the
    mean of the voiced values of each 20 ms unit -- list of lists of Hz, 0.0 = unvoiced."""
    from dissc_amd.f0 import YaaptTracker, f0_per_unit
    if args.device not in _TRACKER:
        _TRACKER[args.device] = YaaptTracker(device=args.device)
    tracks = _TRACKER[args.device]([wav[k, :int(ns[k])] for k in range(len(ns))])
    return [[float(v) for v in f0_per_unit(t, int(T)).astype(np.float32)] for t, T in zip(tracks, frames)]


def partial_fingerprint(args):
    """What a `<out_file>.partial` left by a dead run must agree on before its lines are reused: the input directory, the
    model / quantiser / vocabulary, the F0 mode and the checkpoint files (name, size, mtime).  A leftover from a run with another
    --base_dir (same file names), another quantiser or another --f0 mode would otherwise be spliced into the new manifest."""
    ck = os.path.abspath(args.checkpoint_dir) if args.checkpoint_dir else None
    ck_files = []
    if ck and os.path.isdir(ck):
        for fn in sorted(os.listdir(ck)):
            st = os.stat(os.path.join(ck, fn))
            ck_files.append([fn, st.st_size, int(st.st_mtime)])
    return {"base_dir": os.path.abspath(args.base_dir), "model_name": args.model_name, "quantizer_name": args.quantizer_name,
            "vocab_size": int(args.vocab_size), "f0": args.f0, "checkpoint_dir": ck, "checkpoint_files": ck_files}


def partial_header(args):
    return (json.dumps({"dissc_encode_partial": 2, "fingerprint": partial_fingerprint(args)}) + "\n").encode()


def main(argv=None):
    tm = _Timing()
    parser = argparse.ArgumentParser()
    parser.add_argument('--model_name', default='hubert-base-ls960', help='Name for pretrained dense model name')
    parser.add_argument('--quantizer_name', default='kmeans', help='Name for quantising the hidden units')
    parser.add_argument('--vocab_size', default=100, type=int, help='number of unique HuBERT clusters to used')
    parser.add_argument('--base_dir', default='ESD/wav/train', help='Input audio file path')
    parser.add_argument('--out_file', default='ESD/hubert100/train.txt', help='Output path')
    parser.add_argument('--device', default='cuda:0', help='Device to run on')
    parser.add_argument('--checkpoint_dir', default=None, help='local directory with the HuBERT / k-means files')
    parser.add_argument('--batch_seconds', default=640.0, type=float, help='audio seconds per GPU batch')
    parser.add_argument('--f0', default='yaapt', choices=['yaapt', 'zeros'],
                        help="'yaapt': track F0 like the reference's encoder does; 'zeros': write an all-unvoiced "
                             "track (only valid for the --pred_pitch flows, which never read it)")
    args = parser.parse_args(argv)

    if args.f0 == 'zeros':
        print("WARNING: data/encode.py --f0 zeros writes an all-zero (fully unvoiced) 'f0' track. It is only valid "
              "as input to `infer.py --pred_pitch` (which predicts F0 and never reads it); data/prep_dataset.py "
              "and resynthesis of the source pitch need --f0 yaapt.", file=sys.stderr)
    from dissc_amd.harness import limit_host_threads
    limit_host_threads()  # torch's host pool: a few threads, not one per logical CPU (start-up cost, harness.py)
    from dissc_amd.hubert import SpeechEncoder
    encoder = SpeechEncoder.by_name(dense_model_name=args.model_name, quantizer_model_name=args.quantizer_name,
                                    vocab_size=args.vocab_size, deduplicate=False,
                                    checkpoint_dir=args.checkpoint_dir).to(args.device)
    os.makedirs(Path(args.out_file).parent.absolute(), exist_ok=True)
    tm.add('encoder_load')
    files = os.listdir(args.base_dir)
    # Pass 1: sample counts only (wav headers), so the batches can be formed without holding every
    # waveform in memory; pass 2 loads one batch at a time.
    lengths = {}
    for f in files:
        n = wav_frames(os.path.join(args.base_dir, f))
        if n < 400:
            print(f"\nProblem encoding sample {f}: shorter than one HuBERT frame")
            continue
        lengths[f] = n
    order = sorted(lengths, key=lambda f: (-lengths[f], f))
    # Batches run longest first, but the manifest keeps the reference's line order (os.listdir order, one line per
    # file: data/encode.py:24-41 there) -- positional consumers (`infer.py -n N`, data_split) must see the same
    # subset.  Finished batches are appended to `<out_file>.partial` as they complete; only each file's byte range in
    # it is kept in memory, and the ordered lines are streamed from it into out_file at the end (appended, as the
    # reference's per-line 'a+' does).  A run that died leaves the .partial behind: the next run over the same
    # directory resumes from it (files whose line is already there are not encoded again).
    partial = str(args.out_file) + '.partial'
    commit = partial + '.commit'  # exists while the ordered lines are being appended to out_file: {"out_size_before": N}
    if os.path.exists(commit):
        # the previous run died INSIDE the final append: undo the torn append, so that the rerun cannot duplicate lines -- unless the
        # append had been completed and fsynced (commit says "done": only the clean-up was missing; rolling that back would
        # re-encode the whole directory for nothing)
        try:
            cj = json.load(open(commit))
            before = int(cj["out_size_before"])
            if cj.get("done") and os.path.exists(args.out_file) and os.path.getsize(args.out_file) == int(cj.get("out_size_after", -1)):
                for leftover in (partial, commit):
                    if os.path.exists(leftover):
                        os.remove(leftover)
                print(f"{args.out_file}: the previous run had finished its append ({cj['out_size_after']} bytes); cleaned up, nothing to do")
                return
            if os.path.exists(args.out_file) and os.path.getsize(args.out_file) > before:
                with open(args.out_file, 'r+b') as fo:
                    fo.truncate(before)
                print(f"{args.out_file}: rolled back an interrupted append to {before} bytes")
        except (ValueError, KeyError, OSError) as e:
            print(f"ignoring unreadable {commit}: {e}")
        os.remove(commit)
    where = {}  # file -> (offset, length) of its line in .partial
    header = partial_header(args)
    from dissc_amd import lib as _lib
    if os.path.exists(partial):
        good = 0
        with open(partial, 'rb') as fi:
            first = fi.readline()
            if first != header:
                print(f"discarding {partial}: it was written by a run with a different configuration (or an older version)")
            else:
                good = len(first)
                while True:
                    off, raw = fi.tell(), fi.readline()
                    if not raw:
                        break
                    # a .partial line is "<decoded sample count>\t<manifest line>": the count the batch was really encoded from (a
                    # header that disagrees with its data must not make the line look foreign on every resume)
                    try:
                        head, _, body = raw.partition(b"\t")
                        n_dec = int(head)
                        d = json.loads(body) if body.endswith(b"\n") else None
                        name = d["audio"] if d is not None else None
                    except (ValueError, KeyError, TypeError):
                        name = None
                    if name is None:  # a torn last line: cut it off
                        break
                    # (a line whose unit count does not fit the samples it was encoded from is not a whole line: encode the file again)
                    if name in lengths and len(d.get("units", ())) == _lib.dissc_hubert_frames(n_dec):
                        where[name] = (off + len(head) + 1, len(body))
                    good = off + len(raw)
        if good == 0:
            os.remove(partial)
        else:
            with open(partial, 'r+b') as fo:
                fo.truncate(good)
        if where:
            print(f"resuming from {partial}: {len(where)} of {len(lengths)} files already encoded")
    if not os.path.exists(partial):
        with open(partial, 'wb') as fo:
            fo.write(header)
    order = [f for f in order if f not in where]
    tm.add('scan_headers')
    i = 0
    while i < len(order):
        n0 = lengths[order[i]]
        bsz = max(1, int(args.batch_seconds * 16000 // n0))
        batch = order[i:i + bsz]
        i += len(batch)
        xs = [load_wav(os.path.join(args.base_dir, f))[0] for f in batch]
        ns = np.array([len(x) for x in xs], dtype=np.int32)
        # rows are sized from the DECODED lengths: a header that disagrees with its data must not break the batch
        wav = np.zeros((len(batch), int(ns.max())), dtype=np.float32)
        for k, x in enumerate(xs):
            wav[k, :len(x)] = x
        del xs
        tm.add('read_wavs')
        out = encoder.model(torch.from_numpy(wav), n_samples=torch.from_numpy(ns), want_dense=False)
        units = out["units"].cpu()
        tm.add('hubert_units')
        f0s = track_f0(wav, ns, [int(t) for t in out["frames"]], args) if args.f0 == 'yaapt' else None
        tm.add('f0')
        with open(partial, 'ab') as fo:
            for k, f in enumerate(batch):
                T = int(out["frames"][k])
                u = units[k, :T].tolist()
                f0 = f0s[k] if f0s is not None else [0.0] * T
                raw = (json.dumps({"units": u, "f0": f0, "durations": [1] * T, "audio": f}) + "\n").encode()
                pre = b"%d\t" % int(ns[k])
                where[f] = (fo.tell() + len(pre), len(raw))
                fo.write(pre + raw)
        tm.add('json_lines')
    if where:
        # appended like the reference's per-line 'a+' -- but a crash inside this phase must not leave half an append that the
        # rerun would repeat: the size before the append is recorded first (see `commit` above)
        before = os.path.getsize(args.out_file) if os.path.exists(args.out_file) else 0
        with open(commit, 'w') as fc:
            json.dump({"out_size_before": before}, fc)
            fc.flush()
            os.fsync(fc.fileno())
        with open(args.out_file, 'ab') as fo, open(partial, 'rb') as fi:
            for f in files:
                if f in where:
                    fi.seek(where[f][0])
                    fo.write(fi.read(where[f][1]))
            fo.flush()
            os.fsync(fo.fileno())
        with open(commit, 'w') as fc:  # the append is complete and on disk: a crash from here on must not roll it back
            json.dump({"out_size_before": before, "done": True, "out_size_after": os.path.getsize(args.out_file)}, fc)
            fc.flush()
            os.fsync(fc.fileno())
    if os.path.exists(partial):
        os.remove(partial)
    if os.path.exists(commit):
        os.remove(commit)
    tm.add('ordered_manifest')
    tm.dump(files=len(where))


if __name__ == '__main__':
    main()
