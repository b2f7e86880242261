#!/usr/bin/env python
"""Prepare an encoded dataset for infer.py: optional train/val split, then the per-speaker F0
statistics file that infer.py reads through --f0_path.

Command line compatible with the reference's data/prep_dataset.py (reference
data/prep_dataset.py:6-21): --encoded_path, --stats_path, --seed, --split_method, same defaults.
The statistics are reduced on the GPU (dissc_amd.stats -> csrc/pitch_stats.hip).
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def build_parser():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--encoded_path", default="ESD/hubert100/train.txt",
                    help="units JSONL written by data/encode.py")
    ap.add_argument("--stats_path", default="ESD/hubert100/f0_stats.pkl",
                    help="where to write {speaker: {'mean', 'std'}} of the training part")
    ap.add_argument("--seed", default=42, type=int, help="seed of the random split")
    ap.add_argument("--split_method", default=None,
                    help="'random' or 'paired_val'; omit to treat the whole file as training data")
    ap.add_argument("--device", default="cuda:0", help="GPU that reduces the statistics (extension)")
    ap.add_argument("--allow_unvoiced", action="store_true",
                    help="(extension) pickle NaN statistics for a speaker without voiced frames like the "
                         "reference does, instead of failing")
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    from data_utils import calculate_pitch_stats, data_split
    from infer import seed_everything
    if args.seed is not None:
        seed_everything(args.seed)
    train_path = args.encoded_path
    if args.split_method:
        train_path, _ = data_split(args.encoded_path, split_method=args.split_method)
    calculate_pitch_stats(train_path, args.stats_path, device=args.device, allow_unvoiced=args.allow_unvoiced)


if __name__ == "__main__":
    main()
