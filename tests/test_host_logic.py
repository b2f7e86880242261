"""CPU tests of the host-side logic of the entry points (no GPU): manifest parsing, stats
tensors, morph_seq_len (reference utils.py:39-52), WAV/GT handling of sr/inference.py."""
import importlib.util
import json
import os
import pickle

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_morph_seq_len_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "pred.npz"))
    infer = _load("infer_host", "infer.py")
    for j in range(4):
        got = infer.morph_seq_len(g[f"morph/{j}/units"], g[f"morph/{j}/pitch"], g[f"morph/{j}/lens"])
        np.testing.assert_array_equal(np.asarray(got, dtype=np.float64), g[f"morph/{j}/out"])


def test_manifest_parsing_and_stats(tmp_path, golden_dir):
    from dissc_amd import formats
    p = tmp_path / "m.txt"
    p.write_text(json.dumps({"units": [1, 1, 2], "f0": [0.0, 101.5, float("nan")], "audio": "p225_001.wav"}) + "\n"
                 + "{'units': [3], 'f0': [1.0], 'durations': [1], 'audio': 'x/p226_002.wav'}\n"
                 + "plain/path.wav\n\n")
    m = formats.read_manifest(str(p))
    assert m[0]["units"] == [1, 1, 2] and np.isnan(m[0]["f0"][2])
    assert m[1]["audio"] == "x/p226_002.wav" and m[2] == {"audio": "plain/path.wav"}
    assert formats.speaker_of("a/b/p226_002.wav") == "p226"
    ids = pickle.load(open(os.path.join(golden_dir, "vctk_id_to_spkr.pkl"), "rb"))
    d = formats.spk_id_dict_from_list(ids)
    assert d["p225"] == 0 and d["p231"] == 6 and len(d) == 108
    infer = _load("infer_host2", "infer.py")
    stats = pickle.load(open(os.path.join(golden_dir, "vctk_f0_stats.pkl"), "rb"))
    mean, std = infer.prep_stats_tensors(d, stats)
    assert mean.shape == (108,) and abs(float(mean[d["p330"]]) - 186.08009) < 1e-3
    assert abs(float(std[d["p277"]]) - 29.187865) < 1e-3


def test_sr_inference_gt_and_selection(tmp_path, golden_dir):
    sr = _load("sr_host", "sr/inference.py")
    g = np.load(os.path.join(golden_dir, "sr_inference.npz"))
    gt = sr.load_gt(os.path.join(golden_dir, "s1_1.wav"), 90)
    assert gt.shape == (31950,)  # 32000 samples trimmed to 90 * (32000 // 90)
    np.testing.assert_array_equal(sr.peak_normalize(gt), g["sr/out/p226_001_gt.wav"])
    assert sr.load_gt(str(tmp_path / "missing.wav"), 10) is None
    assert sr.scan_checkpoint(str(tmp_path), "g_") == ""
    for n in ("g_00000002", "g_00000010", "g_00000009"):
        (tmp_path / n).write_bytes(b"")
    assert sr.scan_checkpoint(str(tmp_path), "g_").endswith("g_00000010")
    x = sr.peak_normalize(np.array([0.0, -2.0, 1.0], dtype=np.float32))
    np.testing.assert_array_equal(x, np.array([0.0, -1.0, 0.5], dtype=np.float32))


def test_infer_cli_argument_contract():
    infer = _load("infer_host3", "infer.py")
    import pytest
    with pytest.raises(AssertionError):  # must convert pitch or rhythm (reference infer.py:197)
        infer.main(["--input_path", "x.txt"])
    with pytest.raises(AssertionError):  # wild samples need both (reference infer.py:198)
        infer.main(["--input_path", "x.txt", "--wild_sample", "--pred_len"])


def test_hubert_loader_accepts_fairseq_and_hf_layouts():
    """Checkpoint plumbing of the unit encoder on the host: a fairseq-structured checkpoint with the
    pre-training extras and 12 layers, and the HuggingFace key layout, select the same 6-layer tensors."""
    import torch
    import synthdata as synth
    from dissc_amd.hubert import HubertEncoder
    from oracle import hubert_ref as hr
    sd6 = synth.synth_hubert_state_dict(6)
    sd = dict(synth.synth_hubert_state_dict(12))
    sd.update({k: v for k, v in sd6.items()})
    sd["mask_emb"] = torch.zeros(768)
    sd["final_proj.weight"] = torch.zeros(256, 768)
    sd["label_embs_concat"] = torch.zeros(504, 256)
    a = HubertEncoder({"args": None, "cfg": {}, "model": sd}, None, n_layers=6)._tensors()
    b = HubertEncoder(sd6, None, n_layers=6)._tensors()
    c = HubertEncoder(hr.fairseq_to_hf(sd6), None, n_layers=6)._tensors()
    assert sorted(a) == sorted(b) == sorted(c)
    assert not any(k.startswith(("mask_emb", "final_proj", "label_embs")) for k in a)
    assert max(int(k.split(".")[2]) for k in a if k.startswith("encoder.layers.")) == 5
    assert tuple(a["encoder.pos_conv.0.weight"].shape) == (768, 48, 128)
    for k in a:
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k


def test_parse_speaker_modes_and_unknown_source_speaker():
    """reference sr/dataset.py:132-147 (the five `multispkr` rules) and :319-322 (an unknown speaker is a KeyError)."""
    import argparse
    from pathlib import Path

    import pytest
    from dissc_amd import AttrDict, formats
    p = "/data/corpus/spkA/sess3/p226_001_mic2.wav"
    assert formats.parse_speaker(p, "parent_name") == "sess3"
    assert formats.parse_speaker(Path(p), "parent_parent_name") == "spkA"
    assert formats.parse_speaker(p, "_") == "p226"
    assert formats.parse_speaker(p, "single") == "A"
    assert formats.parse_speaker(p, lambda q: q.stem.upper()) == "P226_001_MIC2"
    with pytest.raises(NotImplementedError):
        formats.parse_speaker(p, "basename")
    with pytest.raises(KeyError, match="p999"):
        formats.speaker_id("p999", {"p226": 0})

    sr = _load("sr_host_spk", "sr/inference.py")
    samples = [{"audio": "x/p226_001.wav", "units": [1, 2, 3], "f0": [0.0, 1.0, 2.0]}]

    def args(**kw):
        d = dict(sample_df=None, target_speakers=["p231"], data_path="/data/spkB", debug=False, n=-1, parts=False,
                 eval_mode=True, pad=None, unseen_speaker=False, vc=True)
        d.update(kw)
        return argparse.Namespace(**d)

    ids = ["p226", "p231", "spkB", "A"]
    for rule, want in (("_", 0), ("parent_name", 2), ("single", 3)):
        jobs, _ = sr.build_jobs(args(), AttrDict({"multispkr": rule}), samples, ids, None, None)
        assert [j["spkr"] for j in jobs] == [want, 1] and jobs[1]["out"] == "p226_001_1_gen.wav"
    # an unknown source speaker: KeyError like the reference's data set, speaker 0 only with --unseen_speaker
    with pytest.raises(KeyError, match="p226"):
        sr.build_jobs(args(), AttrDict({"multispkr": "_"}), samples, ["p231", "p232"], None, None)
    jobs, _ = sr.build_jobs(args(unseen_speaker=True), AttrDict({"multispkr": "_"}), samples, ["p231", "p232"], None, None)
    assert [(j["spkr"], j["out"]) for j in jobs] == [(0, "p226_001_0_gen.wav")]
    with pytest.raises(NotImplementedError):
        sr.build_jobs(args(), AttrDict({"multispkr": "nope"}), samples, ids, None, None)


def test_write_wav_is_scipy_byte_for_byte(tmp_path):
    """dissc_amd.formats.write_wav (what sr/inference.py and convert.py write their outputs with, without importing scipy.io at
    start-up) == scipy.io.wavfile.write -- the call the reference writes with (sr/inference.py:206,250) -- byte for byte"""
    import numpy as np
    from scipy.io import wavfile
    from dissc_amd import formats
    rs = np.random.RandomState(0)
    cases = [rs.standard_normal(n).astype(np.float32) for n in (0, 1, 2, 3, 31999, 32000)]
    cases += [(rs.standard_normal(n) * 3000).astype(np.int16) for n in (0, 1, 7, 16000)]
    cases += [rs.standard_normal((100, 2)).astype(np.float32), np.asarray(rs.standard_normal(50), dtype='>f4')]
    for i, x in enumerate(cases):
        a, b = str(tmp_path / f"a{i}.wav"), str(tmp_path / f"b{i}.wav")
        wavfile.write(a, 16000, x)
        formats.write_wav(b, 16000, x)
        assert open(a, "rb").read() == open(b, "rb").read(), (i, x.dtype, x.shape)
        if x.size:
            rate, back = wavfile.read(b)
            assert rate == 16000 and np.array_equal(back, x.astype(x.dtype.newbyteorder('=')))
    import pytest
    with pytest.raises(ValueError):
        formats.write_wav(str(tmp_path / "c.wav"), 16000, np.zeros(4, np.float64))
