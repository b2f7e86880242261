"""On-disk formats of the DISSC pipeline (host logic).

* units JSONL: one dict per line, keys units:[int], f0:[float], durations (ignored),
  audio:str.  The reference parses lines with ``eval()`` (infer.py:150-151,
  dataset/pitch_dataset.py:27, sr/dataset.py:113-114); we accept the same lines with
  ``json`` first and ``ast.literal_eval`` as the fallback (no arbitrary code execution).
* id_to_spkr.pkl: pickled list[str]; f0_stats.pkl: pickled {spk: {'mean','std'}}.
"""
import ast
import json
import pickle


def parse_line(line):
    line = line.strip()
    try:
        return json.loads(line)  # what the reference writes (json.dumps; NaN/Infinity accepted)
    except ValueError:
        return ast.literal_eval(line)  # python-literal dicts (single quotes)


def read_manifest(path):
    """-> list of dicts for every '{...}' line (other lines are bare audio paths: returned as
    {'audio': line})."""
    out = []
    with open(path) as f:
        for line in f:
            if not line.strip():
                continue
            if line.lstrip()[0] == "{":
                out.append(parse_line(line))
            else:
                out.append({"audio": line.strip()})
    return out


def load_pickle(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def spk_id_dict_from_list(id_to_spkr):
    """reference infer.py:53-54: {name: index}"""
    return {v: k for k, v in enumerate(id_to_spkr)}


def speaker_of(audio_name):
    """speaker = filename prefix before the first '_' (reference dataset/pitch_dataset.py:28,
    sr/dataset.py:139 method '_')."""
    return audio_name.split("/")[-1].split("_")[0]
