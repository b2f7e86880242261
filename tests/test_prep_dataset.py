"""data/prep_dataset.py path: per-speaker F0 statistics and the train/val split,
against outputs of the reference's own data/data_utils.py (tests/golden/prep_expected.pkl)."""
import json
import os
import pickle
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "data"))


@pytest.fixture()
def golden(golden_dir):
    with open(os.path.join(golden_dir, "prep_expected.pkl"), "rb") as f:
        return pickle.load(f)


def test_oracle_pitch_stats_matches_reference(golden, golden_dir):
    from dissc_amd.formats import read_manifest
    from oracle import stats_ref
    got = stats_ref.pitch_stats(read_manifest(os.path.join(golden_dir, "prep_units.txt")))
    assert list(got) == list(golden["stats"])  # same speakers, same (first-seen) order
    for k, v in golden["stats"].items():
        assert got[k]["mean"] == v["mean"] and got[k]["std"] == v["std"]


@pytest.mark.parametrize("method", ["random", "paired_val"])
def test_data_split_matches_reference(golden, golden_dir, tmp_path, method):
    import data_utils
    man = tmp_path / "all.txt"
    shutil.copy(os.path.join(golden_dir, "prep_units.txt"), man)
    np.random.seed(42)
    tr, va = data_utils.data_split(str(man), method)
    assert (open(tr).read(), open(va).read()) == golden["split"][method]
    with pytest.raises(ValueError):
        data_utils.data_split(str(man), "nope")


@pytest.mark.gpu
def test_pitch_stats_kernel_matches_reference(golden, golden_dir, tmp_path):
    import data_utils
    out = tmp_path / "stats.pkl"
    data_utils.calculate_pitch_stats(os.path.join(golden_dir, "prep_units.txt"), str(out))
    got = pickle.load(open(out, "rb"))
    assert list(got) == list(golden["stats"])
    for k, v in golden["stats"].items():
        assert isinstance(got[k]["mean"], np.float64) and isinstance(got[k]["std"], np.float64)
        # fp64 with a different (fixed) summation tree than numpy's pairwise sum
        assert abs(got[k]["mean"] - v["mean"]) <= 1e-12 * abs(v["mean"])
        assert abs(got[k]["std"] - v["std"]) <= 1e-11 * abs(v["std"])


@pytest.mark.gpu
def test_pitch_stats_edge_cases():
    from dissc_amd.stats import pitch_stats
    assert pitch_stats({}) == {}
    rs = np.random.RandomState(0)
    big = (150 + 30 * rs.randn(1_000_003)).astype(np.float32).astype(np.float64)
    big[rs.rand(big.size) < 0.4] = 0.0
    got = pitch_stats({"a": big, "silent": [0.0, 0.0], "one": [0.0, 123.5], "empty": []})
    v = big[big != 0]
    assert abs(got["a"]["mean"] - v.mean()) <= 1e-12 * v.mean()
    assert abs(got["a"]["std"] - v.std()) <= 1e-11 * v.std()
    assert np.isnan(got["silent"]["mean"]) and np.isnan(got["empty"]["std"])  # numpy gives NaN too
    assert got["one"]["mean"] == 123.5 and got["one"]["std"] == 0.0
    again = pitch_stats({"a": big})
    assert again["a"]["mean"] == got["a"]["mean"] and again["a"]["std"] == got["a"]["std"]  # deterministic


@pytest.mark.gpu
def test_prep_dataset_cli(golden, golden_dir, tmp_path):
    man = tmp_path / "train_all.txt"
    shutil.copy(os.path.join(golden_dir, "prep_units.txt"), man)
    stats = tmp_path / "f0_stats.pkl"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "data", "prep_dataset.py"), "--encoded_path", str(man),
                        "--stats_path", str(stats), "--split_method", "random"], capture_output=True, text=True,
                       timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    # seed 42 -> the reference's split; statistics are those of the TRAIN part only
    assert (open(tmp_path / "train.txt").read(), open(tmp_path / "val.txt").read()) == golden["split"]["random"]
    from dissc_amd.formats import read_manifest
    from oracle import stats_ref
    want = stats_ref.pitch_stats(read_manifest(str(tmp_path / "train.txt")))
    got = pickle.load(open(stats, "rb"))
    assert list(got) == list(want)
    for k in want:
        assert abs(got[k]["mean"] - want[k]["mean"]) <= 1e-12 * abs(want[k]["mean"])
        assert abs(got[k]["std"] - want[k]["std"]) <= 1e-11 * abs(want[k]["std"])


@pytest.mark.gpu
def test_all_unvoiced_speaker_fails_loudly(tmp_path):
    """An all-zero F0 track (what `data/encode.py --f0 zeros` writes) must not become NaN statistics
    silently: the CLI fails and writes nothing; --allow_unvoiced restores the reference's NaN pickle."""
    man = tmp_path / "train.txt"
    man.write_text(json.dumps({"units": [1, 2, 3], "f0": [0.0, 0.0, 0.0], "audio": "p1_001.wav"}) + "\n" +
                   json.dumps({"units": [1, 2], "f0": [0.0, 120.0], "audio": "p2_001.wav"}) + "\n")
    stats = tmp_path / "f0_stats.pkl"
    cmd = [sys.executable, os.path.join(ROOT, "data", "prep_dataset.py"), "--encoded_path", str(man),
           "--stats_path", str(stats)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode != 0 and "no voiced F0 frame" in r.stderr and "p1" in r.stderr
    assert not stats.exists()
    r = subprocess.run(cmd + ["--allow_unvoiced"], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    got = pickle.load(open(stats, "rb"))
    assert np.isnan(got["p1"]["mean"]) and got["p2"]["mean"] == 120.0
