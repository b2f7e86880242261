"""Dataset preparation helpers between data/encode.py and infer.py -- same functions, arguments
and file formats as the reference's data/data_utils.py; the statistics themselves are reduced on
the GPU (dissc_amd.stats.pitch_stats -> csrc/pitch_stats.hip)."""
import os
import pickle
import sys
from pathlib import Path

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from dissc_amd.formats import parse_line  # noqa: E402


def data_split(data_path, split_method="random", train_size=.7):
    """Write train.txt / val.txt next to ``data_path`` and return their paths (reference
    data/data_utils.py:8-31).  'random': one ``np.random.rand()`` draw per line, in file order
    (seed with utils.seed_everything first, as data/prep_dataset.py does); 'paired_val': utterance
    numbers <= 24 (``<speaker>_<number>.wav``) go to validation."""
    base = Path(data_path).parent.absolute()
    tr_path, val_path = base / "train.txt", base / "val.txt"
    if split_method not in ("random", "paired_val"):
        raise ValueError(f"Unsupported train-val split method {split_method}")
    with open(data_path) as f, open(tr_path, "w") as f_tr, open(val_path, "w") as f_val:
        for line in f.readlines():
            if split_method == "random":
                to_train = np.random.rand() <= train_size
            else:
                number = int(parse_line(line)["audio"].split("_")[1].split(".")[0])
                to_train = number > 24
            (f_tr if to_train else f_val).write(line)
    return tr_path, val_path


def collect_speaker_f0(data_path):
    """{speaker: [f0 of every frame of every utterance, file order]}; the speaker is the text
    before the first '_' of the 'audio' field, exactly as the reference keys it."""
    by_spk = {}
    with open(data_path) as f:
        for line in f.readlines():
            if not line.strip():
                continue
            d = parse_line(line)
            by_spk.setdefault(d["audio"].split("_")[0], []).extend(d["f0"])
    return by_spk


def calculate_pitch_stats(data_path, out_path, device="cuda:0", allow_unvoiced=False):
    """{speaker: {'mean', 'std'}} of the voiced frames (f0 != 0), pickled to ``out_path``
    (reference data/data_utils.py:33-46; np.float64 scalars, population std).
    A speaker without any voiced frame raises (nothing is written) unless ``allow_unvoiced``:
    the reference would pickle NaN statistics that poison every later stage silently."""
    from dissc_amd.stats import pitch_stats
    stats = pitch_stats(collect_speaker_f0(data_path), device=device,
                        on_unvoiced="nan" if allow_unvoiced else "raise")
    with open(out_path, "wb") as f_out:
        pickle.dump(stats, f_out)
