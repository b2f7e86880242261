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


def parse_speaker(path, method):
    """Speaker name of an audio path under the vocoder config's ``multispkr`` rule (reference
    sr/dataset.py:132-147): 'parent_name' = the directory holding the file, 'parent_parent_name' = the one
    above it, '_' = filename prefix before the first underscore, 'single' = the constant 'A', or a callable
    taking the ``pathlib.Path``.  Anything else raises NotImplementedError like the reference."""
    from pathlib import Path
    p = Path(path) if isinstance(path, str) else path
    if method == "parent_name":
        return p.parent.name
    if method == "parent_parent_name":
        return p.parent.parent.name
    if method == "_":
        return p.name.split("_")[0]
    if method == "single":
        return "A"
    if callable(method):
        return method(p)
    raise NotImplementedError(f"multispkr = {method!r}: expected 'parent_name', 'parent_parent_name', '_', 'single' "
                              "or a callable (reference sr/dataset.py:132-147)")


def speaker_id(name, spkr_to_id, what="source"):
    """Index of a speaker in ``id_to_spkr``; an absent name raises KeyError as the reference's
    ``self.spkr_to_id[spkr_name]`` does (sr/dataset.py:319-322) instead of silently conditioning on speaker 0."""
    try:
        return int(spkr_to_id[name])
    except KeyError:
        raise KeyError(f"{what} speaker {name!r} is not in id_to_spkr ({len(spkr_to_id)} speakers); "
                       "pass --unseen_speaker for speakers outside the training set") from None


def write_wav(path, rate, data):
    """``scipy.io.wavfile.write(path, rate, data)`` for the two sample types this repo writes -- float32 (IEEE-float WAVE:
    18-byte fmt chunk + fact chunk) and int16 (PCM) --, byte for byte (tests/test_host_logic.py compares the files), without
    importing scipy.io (0.25-0.6 s of a CLI's start-up; the reference writes through the same scipy call: sr/inference.py:206,250)."""
    import struct

    import numpy as np
    data = np.asarray(data)
    if data.dtype.byteorder == '>':
        data = data.astype(data.dtype.newbyteorder('<'))
    if data.dtype == np.float32:
        tag = 3
    elif data.dtype == np.int16:
        tag = 1
    else:
        raise ValueError(f"write_wav: unsupported dtype {data.dtype}")
    channels = 1 if data.ndim == 1 else data.shape[1]
    bits = data.dtype.itemsize * 8
    fmt = struct.pack('<HHIIHH', tag, channels, int(rate), int(rate) * (bits // 8) * channels, channels * (bits // 8), bits)
    if tag != 1:
        fmt += b'\x00\x00'  # cbSize of a non-PCM format
    body = b'WAVE' + b'fmt ' + struct.pack('<I', len(fmt)) + fmt
    if tag != 1:
        body += b'fact' + struct.pack('<II', 4, data.shape[0])
    payload = np.ascontiguousarray(data)
    if len(body) + 8 + payload.nbytes + 4 > 0xFFFFFFFF:
        raise ValueError("write_wav: data exceeds the WAV size limit")
    body += b'data' + struct.pack('<I', payload.nbytes)
    with open(path, 'wb') as f:
        f.write(b'RIFF' + struct.pack('<I', len(body) + payload.nbytes) + body)
        f.write(payload.data if payload.size else b'')
        # (no pad byte: both sample types have an even size)
