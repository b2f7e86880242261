#!/usr/bin/env python
"""Train/val split + per-speaker F0 statistics of an encoded dataset: same command line as the
reference's data/prep_dataset.py (reference data/prep_dataset.py:6-21).  The statistics file it
writes is what infer.py reads through --f0_path."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

from data_utils import calculate_pitch_stats, data_split  # noqa: E402
from infer import seed_everything  # noqa: E402


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--encoded_path', default='ESD/hubert100/train.txt', help='Path for HuBERT encodings')
    parser.add_argument('--stats_path', default='ESD/hubert100/f0_stats.pkl', help='Output path for train speaker stats')
    parser.add_argument('--seed', default=42, type=int, help='number of unique HuBERT clusters to used')
    parser.add_argument('--split_method', default=None, help='Method for train-test split. If None encoded path is all train and no split is performed')
    parser.add_argument('--device', default='cuda:0', help='GPU that reduces the statistics (not in the reference)')

    args = parser.parse_args()

    if args.seed is not None:
        seed_everything(args.seed)
    if args.split_method:
        train_path, _ = data_split(args.encoded_path, split_method=args.split_method)
    else:
        train_path = args.encoded_path
    calculate_pitch_stats(train_path, args.stats_path, device=args.device)
