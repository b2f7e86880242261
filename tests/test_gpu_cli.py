"""End-to-end entry points on the GPU vs outputs of the reference's own CLI code
(tests/golden/pred.npz 'val/*' = reference infer.infer(); sr_inference.npz = reference
sr/inference.py init_worker + inference)."""
import importlib.util
import json
import os
import shutil
import sys

import numpy as np
import pytest
import torch
from scipy.io import wavfile

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_infer_cli_matches_reference(gpu, golden_dir, tmp_path):
    import synthdata as synth
    g = np.load(os.path.join(golden_dir, "pred.npz"))
    td = str(tmp_path)
    for d in ("len", "pitch", "out", "in"):
        os.makedirs(f"{td}/{d}")
    torch.save(synth.synth_len_state_dict(100, 108), f"{td}/len/best_model.pth")
    torch.save(synth.synth_len_norm_stats(), f"{td}/len/len_norm_stats.pth")
    torch.save(synth.synth_pitch_state_dict("new", 100, 108), f"{td}/pitch/best_model.pth")
    shutil.copy(os.path.join(golden_dir, "vctk_id_to_spkr.pkl"), f"{td}/in/id_to_spkr.pkl")
    open(f"{td}/in/val.txt", "w").write(str(g["val/manifest"]))
    infer = _load("dissc_infer_cli", "infer.py")
    infer.main(["--input_path", f"{td}/in/val.txt", "-n", "1000", "--out_path", f"{td}/out", "--pred_len",
                "--pred_pitch", "--len_model", f"{td}/len/", "--f0_model", f"{td}/pitch/", "--f0_model_type",
                "new", "--f0_path", os.path.join(golden_dir, "vctk_f0_stats.pkl"), "--vc",
                "--target_speakers", "p231", "p225", "--device", "cuda:0"])
    assert sorted(os.listdir(f"{td}/out")) == ["p225_val.txt", "p231_val.txt", "val.txt"]
    for fn in os.listdir(f"{td}/out"):
        lines = open(f"{td}/out/{fn}").read().strip().split("\n")
        assert len(lines) == 3
        for i, ln in enumerate(lines):
            d = json.loads(ln)
            assert d["audio"] == str(g[f"val/{fn}/{i}/audio"])
            np.testing.assert_array_equal(d["units"], g[f"val/{fn}/{i}/units"])  # exact
            want = g[f"val/{fn}/{i}/f0"]
            got = np.array(d["f0"])
            flips = (got == 0) != (want == 0)
            assert flips.sum() <= 1
            assert np.abs(got[~flips] - want[~flips]).max() <= 2e-5


def _sr_inference_setup(td, golden_dir, g):
    """checkpoint dir, wav dir and manifest of the golden sr/inference.py case -> CLI arguments"""
    import synthdata as synth
    for d in ("ckpt", "wav", "out", "meta"):
        os.makedirs(f"{td}/{d}")
    cfg = dict(synth.VCTK_CONFIG, input_training_file=f"{td}/meta/train.txt", f0_normalize=False,
               f0_stats=None, test_base_path=f"{td}/wav")
    json.dump(cfg, open(f"{td}/ckpt/config.json", "w"))
    torch.save({"generator": synth.synth_generator_state_dict(seed=0)}, f"{td}/ckpt/g_00000001")
    shutil.copy(os.path.join(golden_dir, "vctk_id_to_spkr.pkl"), f"{td}/meta/id_to_spkr.pkl")
    names = [str(x) for x in g["sr/names"]]
    with open(f"{td}/man.txt", "w") as f:
        for i, nm in enumerate(names):
            shutil.copy(os.path.join(golden_dir, f"s1_{i + 1}.wav"), f"{td}/wav/{nm}")
            f.write(json.dumps({"units": g[f"sr/units{i}"].tolist(), "f0": g[f"sr/f0{i}"].tolist(),
                                "audio": nm}) + "\n")
    return ["--input_code_file", f"{td}/man.txt", "--data_path", f"{td}/wav", "--output_dir", f"{td}/out",
            "--checkpoint_file", f"{td}/ckpt/", "--vc", "--target-speakers", "p231", "p225", "-n", "-1"]


def test_sr_inference_two_ranks_equal_one_process(gpu, golden_dir, tmp_path):
    """The rank-sharded path of sr/inference.py with device tensors: 2 ranks (sharing this box's GPU,
    gloo instead of RCCL) must write exactly the files of a single process."""
    import subprocess
    g = np.load(os.path.join(golden_dir, "sr_inference.npz"))
    td = str(tmp_path)
    args = _sr_inference_setup(td, golden_dir, g)
    cli = _load("dissc_sr_inference_cli2", "sr/inference.py")
    cli.main(args)
    os.rename(f"{td}/out", f"{td}/out_single")
    os.makedirs(f"{td}/out")
    env = dict(os.environ, DISSC_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29537",
                        os.path.join(ROOT, "sr", "inference.py")] + args,
                       env=env, capture_output=True, text=True, timeout=900, cwd=td)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    files = sorted(os.listdir(f"{td}/out_single"))
    assert files and sorted(os.listdir(f"{td}/out")) == files
    for fn in files:
        assert open(f"{td}/out/{fn}", "rb").read() == open(f"{td}/out_single/{fn}", "rb").read(), fn


def _assert_wav_close(data, ref, fn):
    """Generated files are int16-TRUNCATED and then divided by their peak (reference sr/inference.py:73-75,206): a ~1e-6
    difference of the waveform moves a sample by one int16 LSB (1 / peak ~ 3e-5) now and then -- and when that sample is
    the PEAK itself, every sample of the file is rescaled by one part in ~30 000.  So: the best-fit scale between the two
    files may differ from 1 by at most one LSB of the peak; after it, a few per cent of the samples differ by one LSB,
    everything else is bit-identical."""
    s = float(np.dot(ref.astype(np.float64), data.astype(np.float64)) / max(np.dot(data.astype(np.float64), data.astype(np.float64)), 1e-30))
    assert abs(s - 1.0) <= 6e-5, (fn, s)
    d = np.abs(data.astype(np.float64) * s - ref)
    assert d.max() <= 1e-4, (fn, d.max())
    assert np.sqrt(np.mean(d ** 2)) <= 1e-5, (fn, np.sqrt(np.mean(d ** 2)))
    assert (d > 1e-6).mean() <= 0.04, (fn, (d > 1e-6).mean())


def _check_wavs(out_dir, g, prefix):
    want_files = sorted(k[len(prefix):] for k in g.files if k.startswith(prefix))
    assert sorted(os.listdir(out_dir)) == want_files
    for fn in want_files:
        rate, data = wavfile.read(f"{out_dir}/{fn}")
        ref = g[prefix + fn]
        assert rate == 16000 and data.dtype == np.float32 and data.shape == ref.shape, fn
        if fn.endswith("_gt.wav"):
            np.testing.assert_array_equal(data, ref)
        else:
            _assert_wav_close(data, ref, fn)


def test_sr_inference_unseen_speaker_f0_stats_parts_and_sample_df(gpu, golden_dir, tmp_path):
    """More of the reference harness's branches, against its own outputs: an unseen source speaker
    (no resynthesis, speaker id 0), --f0-stats (voiced F0 re-normalised to each target, with the
    global fallback for a target that has no entry), --parts output names, and --sample_df (only the
    listed source/target pairs, no ground-truth copies)."""
    import pandas as pd
    import synthdata as synth
    g = np.load(os.path.join(golden_dir, "sr_inference.npz"))
    td = str(tmp_path)
    for d in ("ckpt", "data/wav", "out_b", "out_c", "meta"):
        os.makedirs(f"{td}/{d}")
    cfg = dict(synth.VCTK_CONFIG, input_training_file=f"{td}/meta/train.txt", f0_normalize=False,
               f0_stats=None, test_base_path=f"{td}/data/wav")
    json.dump(cfg, open(f"{td}/ckpt/config.json", "w"))
    torch.save({"generator": synth.synth_generator_state_dict(seed=0)}, f"{td}/ckpt/g_00000001")
    shutil.copy(os.path.join(golden_dir, "vctk_id_to_spkr.pkl"), f"{td}/meta/id_to_spkr.pkl")
    tid, m0, s0, mg, sg = g["srb/stats"]
    torch.save({int(tid): {"f0_mean": float(m0), "f0_std": float(s0)}, "f0_mean": float(mg), "f0_std": float(sg)},
               f"{td}/meta/tgt_f0_stats.pt")
    for key, man, f0key in (("srb/names", "man_b.txt", "srb/f0"), ("src/names", "man_c.txt", "sr/f0")):
        with open(f"{td}/{man}", "w") as f:
            for i, nm in enumerate(str(x) for x in g[key]):
                shutil.copy(os.path.join(golden_dir, f"s1_{i + 1}.wav"), f"{td}/data/wav/{nm}")
                f.write(json.dumps({"units": g[f"srb/units{i}"].tolist(), "f0": g[f"{f0key}{i}"].tolist(),
                                    "audio": nm}) + "\n")
    cli = _load("dissc_sr_inference_cli3", "sr/inference.py")
    cli.main(["--input_code_file", f"{td}/man_b.txt", "--data_path", f"{td}/data/wav", "--output_dir", f"{td}/out_b",
              "--checkpoint_file", f"{td}/ckpt/", "--vc", "--target-speakers", "p231", "p225", "-n", "-1",
              "--f0-stats", f"{td}/meta/tgt_f0_stats.pt", "--parts", "--unseen_speaker",
              "--id_to_spkr", f"{td}/meta/id_to_spkr.pkl"])
    _check_wavs(f"{td}/out_b", g, "srb/out/")
    pairs = g["src/pairs"]
    pd.DataFrame({"syn_sample": list(pairs[0]), "syn_trgt": list(pairs[1])}).to_csv(f"{td}/meta/pairs.csv")
    cli.main(["--input_code_file", f"{td}/man_c.txt", "--data_path", f"{td}/data/wav", "--output_dir", f"{td}/out_c",
              "--checkpoint_file", f"{td}/ckpt/", "--vc", "--target-speakers", "p231", "p225", "p226", "-n", "-1",
              "--sample_df", f"{td}/meta/pairs.csv"])
    _check_wavs(f"{td}/out_c", g, "src/out/")


def test_sr_inference_cli_matches_reference(gpu, golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "sr_inference.npz"))
    td = str(tmp_path)
    cli = _load("dissc_sr_inference_cli", "sr/inference.py")
    cli.main(_sr_inference_setup(td, golden_dir, g))
    want_files = sorted(k[len("sr/out/"):] for k in g.files if k.startswith("sr/out/"))
    assert sorted(os.listdir(f"{td}/out")) == want_files
    for fn in want_files:
        rate, data = wavfile.read(f"{td}/out/{fn}")
        ref = g["sr/out/" + fn]
        assert rate == 16000 and data.dtype == np.float32 and data.shape == ref.shape, fn
        if fn.endswith("_gt.wav"):
            np.testing.assert_array_equal(data, ref)
        else:
            _assert_wav_close(data, ref, fn)


def test_encode_cli_roundtrip(gpu, golden_dir, tmp_path):
    """data/encode.py on the reference's two fixture wavs: units equal the CPU oracle's."""
    from oracle import hubert_ref as hr
    import synthdata as synth
    td = str(tmp_path)
    os.makedirs(f"{td}/ckpt")
    os.makedirs(f"{td}/wav")
    sd = synth.synth_hubert_state_dict(6)
    centers = synth.synth_kmeans_centers()
    torch.save({"model": sd}, f"{td}/ckpt/hubert-base-ls960.pt")
    np.save(f"{td}/ckpt/kmeans_100.npy", centers.numpy())
    for i in (1, 2):
        shutil.copy(os.path.join(golden_dir, f"s1_{i}.wav"), f"{td}/wav/s1_{i}.wav")
    cli = _load("dissc_encode_cli", "data/encode.py")
    cli.main(["--base_dir", f"{td}/wav", "--out_file", f"{td}/out/enc.txt", "--checkpoint_dir", f"{td}/ckpt"])
    lines = [json.loads(x) for x in open(f"{td}/out/enc.txt").read().strip().split("\n")]
    assert sorted(d["audio"] for d in lines) == ["s1_1.wav", "s1_2.wav"]
    for d in lines:
        x, sr = cli.load_wav(f"{td}/wav/{d['audio']}")
        assert sr == 16000 and len(x) == 32000
        units, dense = hr.encode(sd, centers, torch.from_numpy(x)[None])
        assert len(d["units"]) == 99 == len(d["f0"]) == len(d["durations"])
        # the CLI does not return features: the asserted feature-error bound stands in for the measured one
        hr.check_units(np.array(d["units"]), units.numpy(), dense, centers, tag=d["audio"])
    # a run that died leaves <out_file>.partial: the next run resumes from it (the finished file is not encoded again --
    # its line comes back verbatim, a torn last line is dropped) and the manifest keeps the listdir order
    import argparse
    first = lines[0]
    marked = dict(first, durations=[7] * 99)
    ns = argparse.Namespace(base_dir=f"{td}/wav", model_name="hubert-base-ls960", quantizer_name="kmeans", vocab_size=100,
                            f0="yaapt", checkpoint_dir=f"{td}/ckpt")
    with open(f"{td}/out/enc2.txt.partial", "wb") as f:
        f.write(cli.partial_header(ns) + ("32000\t" + json.dumps(marked) + "\n" + '32000\t{"units": [1, 2').encode())  # "<samples>\t<line>"
    cli.main(["--base_dir", f"{td}/wav", "--out_file", f"{td}/out/enc2.txt", "--checkpoint_dir", f"{td}/ckpt"])
    again = [json.loads(x) for x in open(f"{td}/out/enc2.txt").read().strip().split("\n")]
    assert [d["audio"] for d in again] == [d["audio"] for d in lines]
    assert again[0] == marked and again[1] == lines[1]
    assert not os.path.exists(f"{td}/out/enc2.txt.partial")
    # a leftover from a run with ANOTHER configuration (here: --f0 zeros) or whose line does not fit the file (wrong
    # unit count) is not spliced in; and an append that was interrupted is rolled back instead of repeated
    stale = dict(first, durations=[9] * 99)
    with open(f"{td}/out/enc3.txt.partial", "wb") as f:
        f.write(cli.partial_header(argparse.Namespace(**dict(vars(ns), f0="zeros"))) + ("32000\t" + json.dumps(stale) + "\n").encode())
    cli.main(["--base_dir", f"{td}/wav", "--out_file", f"{td}/out/enc3.txt", "--checkpoint_dir", f"{td}/ckpt"])
    assert [json.loads(x) for x in open(f"{td}/out/enc3.txt").read().strip().split("\n")] == lines
    short = dict(first, units=first["units"][:50])
    with open(f"{td}/out/enc4.txt.partial", "wb") as f:
        f.write(cli.partial_header(ns) + ("32000\t" + json.dumps(short) + "\n").encode())
    with open(f"{td}/out/enc4.txt", "w") as f:
        f.write("kept\n" + json.dumps(first)[:40])      # a torn append of a dead run ...
    with open(f"{td}/out/enc4.txt.partial.commit", "w") as f:
        json.dump({"out_size_before": 5}, f)             # ... which began at byte 5
    cli.main(["--base_dir", f"{td}/wav", "--out_file", f"{td}/out/enc4.txt", "--checkpoint_dir", f"{td}/ckpt"])
    got = open(f"{td}/out/enc4.txt").read().strip().split("\n")
    assert got[0] == "kept" and [json.loads(x) for x in got[1:]] == lines
    assert not os.path.exists(f"{td}/out/enc4.txt.partial.commit")
    # (a) the line is validated against the DECODED sample count it was encoded from, stored in front of it -- a wav whose
    # header disagrees with its data is accepted on resume; (b) a run that died after its append was complete and fsynced (commit
    # says "done") is cleaned up, not rolled back and re-encoded
    lying = dict(first, units=first["units"][:50], f0=first["f0"][:50], durations=[3] * 50)   # encoded from 16 320 samples, says the line
    with open(f"{td}/out/enc5.txt.partial", "wb") as f:
        f.write(cli.partial_header(ns) + ("16320\t" + json.dumps(lying) + "\n").encode())
    cli.main(["--base_dir", f"{td}/wav", "--out_file", f"{td}/out/enc5.txt", "--checkpoint_dir", f"{td}/ckpt"])
    got5 = [json.loads(x) for x in open(f"{td}/out/enc5.txt").read().strip().split("\n")]
    assert got5[0] == lying and got5[1] == lines[1]
    done_text = open(f"{td}/out/enc5.txt").read()
    with open(f"{td}/out/enc5.txt.partial", "wb") as f:
        f.write(cli.partial_header(ns))
    with open(f"{td}/out/enc5.txt.partial.commit", "w") as f:
        json.dump({"out_size_before": 0, "done": True, "out_size_after": len(done_text.encode())}, f)
    cli.main(["--base_dir", f"{td}/wav", "--out_file", f"{td}/out/enc5.txt", "--checkpoint_dir", f"{td}/ckpt"])
    assert open(f"{td}/out/enc5.txt").read() == done_text
    assert not os.path.exists(f"{td}/out/enc5.txt.partial") and not os.path.exists(f"{td}/out/enc5.txt.partial.commit")


def test_in_memory_converter_equals_file_pipeline(gpu, golden_dir, tmp_path):
    """dissc_amd.pipeline.Converter (device-resident hand-off) == encode.py -> infer.py ->
    sr/inference.py through JSONL files, sample for sample."""
    import synthdata as synth
    import dissc_amd
    from dissc_amd import predictors as P
    from dissc_amd.hubert import HubertEncoder
    from dissc_amd.pipeline import Converter
    td = str(tmp_path)
    for d in ("hub", "wav", "enc", "len", "pitch", "pred", "ckpt", "meta", "out"):
        os.makedirs(f"{td}/{d}")
    hsd, centers = synth.synth_hubert_state_dict(6), synth.synth_kmeans_centers()
    torch.save({"model": hsd}, f"{td}/hub/hubert-base-ls960.pt")
    np.save(f"{td}/hub/kmeans_100.npy", centers.numpy())
    names = ["p226_001.wav", "p300_002.wav"]
    for i, nm in enumerate(names):
        shutil.copy(os.path.join(golden_dir, f"s1_{i + 1}.wav"), f"{td}/wav/{nm}")
    torch.save(synth.synth_len_state_dict(100, 108), f"{td}/len/best_model.pth")
    torch.save(synth.synth_len_norm_stats(), f"{td}/len/len_norm_stats.pth")
    torch.save(synth.synth_pitch_state_dict("new", 100, 108), f"{td}/pitch/best_model.pth")
    shutil.copy(os.path.join(golden_dir, "vctk_id_to_spkr.pkl"), f"{td}/enc/id_to_spkr.pkl")
    shutil.copy(os.path.join(golden_dir, "vctk_id_to_spkr.pkl"), f"{td}/meta/id_to_spkr.pkl")
    cfg = dict(synth.VCTK_CONFIG, input_training_file=f"{td}/meta/train.txt", f0_normalize=False, f0_stats=None)
    json.dump(cfg, open(f"{td}/ckpt/config.json", "w"))
    gsd = synth.synth_generator_state_dict(seed=0)
    torch.save({"generator": gsd}, f"{td}/ckpt/g_00000001")
    # file pipeline
    _load("enc_cli2", "data/encode.py").main(["--base_dir", f"{td}/wav", "--out_file", f"{td}/enc/val.txt",
                                              "--checkpoint_dir", f"{td}/hub"])
    _load("infer_cli2", "infer.py").main(["--input_path", f"{td}/enc/val.txt", "-n", "10", "--out_path", f"{td}/pred",
                                          "--pred_len", "--pred_pitch", "--len_model", f"{td}/len/", "--f0_model",
                                          f"{td}/pitch/", "--f0_path", os.path.join(golden_dir, "vctk_f0_stats.pkl"),
                                          "--vc", "--target_speakers", "p231"])
    _load("sr_cli2", "sr/inference.py").main(["--input_code_file", f"{td}/pred/p231_val.txt", "--data_path", f"{td}/wav",
                                              "--output_dir", f"{td}/out", "--checkpoint_file", f"{td}/ckpt/", "--vc",
                                              "--target-speakers", "p231", "--unseen_speaker", "--id_to_spkr",
                                              f"{td}/meta/id_to_spkr.pkl", "-n", "-1"])
    # in-memory pipeline with the same models
    enc = HubertEncoder(hsd, centers, 6).to("cuda:0")
    lm = P.LenPredictor(100, 108).to("cuda:0")
    lm.load_state_dict(synth.synth_len_state_dict(100, 108))
    lm.norm_mean, lm.norm_std = synth.synth_len_norm_stats()
    pm = P.PitchPredictor(100, 108).to("cuda:0")
    pm.load_state_dict(synth.synth_pitch_state_dict("new", 100, 108))
    g = dissc_amd.CodeGenerator(cfg).to("cuda:0")
    g.load_state_dict(gsd)
    g.eval().remove_weight_norm()
    cli = _load("enc_cli3", "data/encode.py")
    waves = [cli.load_wav(f"{td}/wav/{nm}")[0] for nm in names]
    out = Converter(enc, lm, pm, g)(waves, [6])  # p231 = id 6
    for i, nm in enumerate(names):
        rate, ref = wavfile.read(f"{td}/out/{nm[:-4]}_6_gen.wav")
        got = out[(i, 6)]
        assert got.shape == ref.shape and rate == 16000
        # the only difference is the JSON text round trip of F0 (float32 -> decimal -> float32: exact)
        np.testing.assert_array_equal(got, ref)


def test_infer_cli_rhythm_only_matches_reference(gpu, golden_dir, tmp_path):
    """--pred_len without --pred_pitch: predicted lengths + the source F0 morphed per unit run."""
    import synthdata as synth
    g = np.load(os.path.join(golden_dir, "pred.npz"))
    td = str(tmp_path)
    for d in ("len", "out", "in"):
        os.makedirs(f"{td}/{d}")
    torch.save(synth.synth_len_state_dict(100, 108), f"{td}/len/best_model.pth")
    torch.save(synth.synth_len_norm_stats(), f"{td}/len/len_norm_stats.pth")
    shutil.copy(os.path.join(golden_dir, "vctk_id_to_spkr.pkl"), f"{td}/in/id_to_spkr.pkl")
    open(f"{td}/in/val.txt", "w").write(str(g["val/manifest"]))
    infer = _load("dissc_infer_cli_len", "infer.py")
    infer.main(["--input_path", f"{td}/in/val.txt", "-n", "3", "--out_path", f"{td}/out", "--pred_len",
                "--len_model", f"{td}/len/", "--f0_path", os.path.join(golden_dir, "vctk_f0_stats.pkl"), "--vc",
                "--target_speakers", "p231", "p225"])
    for fn in ("val.txt", "p231_val.txt", "p225_val.txt"):
        lines = open(f"{td}/out/{fn}").read().strip().split("\n")
        for i, ln in enumerate(lines):
            d = json.loads(ln)
            np.testing.assert_array_equal(d["units"], g[f"lenonly/{fn}/{i}/units"])
            assert np.abs(np.array(d["f0"]) - g[f"lenonly/{fn}/{i}/f0"]).max() <= 1e-5


def test_infer_cli_pitch_only_pairs_in_hz_matches_reference(gpu, golden_dir, tmp_path):
    """--pred_pitch without --pred_len, base pitch model, --norm_pitch given (store_false: F0 written
    in Hz, de-normalised with the TARGET speaker's statistics), --sample_df (only the listed pairs,
    no reconstruction file)."""
    import pandas as pd
    import synthdata as synth
    g = np.load(os.path.join(golden_dir, "pred.npz"))
    td = str(tmp_path)
    for d in ("pitch", "out", "in"):
        os.makedirs(f"{td}/{d}")
    torch.save(synth.synth_pitch_state_dict("base", 100, 108), f"{td}/pitch/best_model.pth")
    shutil.copy(os.path.join(golden_dir, "vctk_id_to_spkr.pkl"), f"{td}/in/id_to_spkr.pkl")
    open(f"{td}/in/val.txt", "w").write(str(g["val/manifest"]))
    pairs = g["pitchonly/pairs"]
    pd.DataFrame({"syn_sample": list(pairs[0]), "syn_trgt": list(pairs[1])}).to_csv(f"{td}/in/pairs.csv")
    infer = _load("dissc_infer_cli_pitch", "infer.py")
    infer.main(["--input_path", f"{td}/in/val.txt", "-n", "3", "--out_path", f"{td}/out", "--pred_pitch",
                "--f0_model", f"{td}/pitch/", "--f0_model_type", "base", "--norm_pitch",
                "--f0_path", os.path.join(golden_dir, "vctk_f0_stats.pkl"), "--vc",
                "--target_speakers", "p231", "p225", "--sample_df", f"{td}/in/pairs.csv"])
    assert sorted(os.listdir(f"{td}/out")) == ["p225_val.txt", "p231_val.txt"]  # no reconstruction file
    for fn in ("p225_val.txt", "p231_val.txt"):
        lines = open(f"{td}/out/{fn}").read().strip().split("\n")
        assert len(lines) == int(g[f"pitchonly/{fn}/n"])
        for i, ln in enumerate(lines):
            d = json.loads(ln)
            assert d["audio"] == str(g[f"pitchonly/{fn}/{i}/audio"])
            np.testing.assert_array_equal(d["units"], g[f"pitchonly/{fn}/{i}/units"])  # units pass through
            want, got = g[f"pitchonly/{fn}/{i}/f0"], np.array(d["f0"])
            flips = (got == 0) != (want == 0)
            assert flips.sum() <= 1
            assert np.abs(got[~flips] - want[~flips]).max() <= 2e-3  # Hz (normalised value x std ~ 30-50)
