#!/usr/bin/env python
"""Prepare audio for data/encode.py: resample to 16 kHz, optionally trim leading / trailing silence, optionally
zero-pad to a multiple of 1280 samples.  Same command line and file layout as the reference's data/preprocess.py
(reference data/preprocess.py:19-57: --srcdir --outdir --trim --pad --postfix; outputs ``outdir/<file name>``).

Resampling runs on the MI355X (dissc_amd.audio.resample -> libdissc_hip.so ``dissc_resample``); files are processed
one after the other in this process instead of a 40-worker pool.  Only WAV input is read (scipy; the reference's
soundfile also reads FLAC etc.).  Parity of the resampler / trim rule with resampy / librosa is unpinned (absent
offline, restated in oracle/preprocess_ref.py); the padding is the reference's own numpy rule.
"""
import argparse
import os
import sys
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pad_data(p, out_dir, trim=False, pad=False, device="cuda:0"):
    from dissc_amd import audio
    data, sr = audio.read_audio(str(p))
    if data.ndim > 1:
        raise ValueError(f"{p}: multi-channel audio is not supported (the reference would resample along axis -1)")
    if sr != 16000:
        data = audio.resample(data, sr, 16000, device=device)
        sr = 16000
    if trim:
        data, _ = audio.trim(data, top_db=20)
    if pad:
        data = audio.pad_to_multiple(data, 1280)
    outpath = Path(out_dir) / Path(p).name
    outpath.parent.mkdir(exist_ok=True, parents=True)
    audio.write_pcm16(str(outpath), data, sr)
    return outpath


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--srcdir', type=Path, required=True)
    parser.add_argument('--outdir', type=Path, required=True)
    parser.add_argument('--trim', action='store_true')
    parser.add_argument('--pad', action='store_true')
    parser.add_argument('--postfix', type=str, default='wav')
    parser.add_argument('--device', default='cuda:0', help='GPU that resamples (extension)')
    args = parser.parse_args(argv)
    files = sorted(Path(args.srcdir).glob(f'**/*{args.postfix}'))
    for p in files:
        pad_data(p, Path(args.outdir), trim=args.trim, pad=args.pad, device=args.device)
    print(f'{len(files)} files written to {args.outdir}')


if __name__ == '__main__':
    main()
