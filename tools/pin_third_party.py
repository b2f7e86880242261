#!/usr/bin/env python
"""Pin the three stages whose arithmetic lives in un-vendored third parties -- in ONE command, wherever those
libraries can be imported (they cannot in the build container: no network).

    python tools/pin_third_party.py --check          # what is importable here, the pins the reference names, what would be written
    python tools/pin_third_party.py [--only hubert,yaapt,resample] [--out tests/golden] [--no-tests] [--gpu]
                                    [--real-encoder] [--checkpoint-dir DIR]

What it writes (each only when its library imports; a missing library is reported and skipped, exit code 0):

  resample_resampy.npz   ``resampy.resample(x, sr, 16000)`` (reference data/preprocess.py:19-24, sr/dataset.py:225-227)
                         and ``librosa.effects.trim(x, top_db=20)`` (data/preprocess.py:26-27) of seeded signals
  yaapt_amfm.npz         ``amfm_decompy.pYAAPT.yaapt`` called exactly like reference sr/dataset.py:27-43 (frame_length 20,
                         frame_space 5, nccf_thresh1 0.25, tda_frame_length 25, 10 ms zero padding, ``samp_values``) on the
                         two speech fixtures and on known-F0 synthetic signals
  hubert_fairseq.npz     fairseq's OWN ``HubertModel.extract_features(output_layer=6)`` (pinned commit dd106d95, reference
                         README.md:31-34) run with this repo's seeded synthetic weights loaded into it (the real
                         checkpoint is a download; the synthetic one travels as a seed) + sklearn k-means predict
  hubert_textless_real.npz  (--real-encoder) textless ``SpeechEncoder.by_name(...)(waveform)`` -- the reference's literal
                         call, data/encode.py:21-22,32 -- on the speech fixtures with the REAL hubert-base-ls960 / km100
                         files; the HIP test for it needs the same files under --checkpoint-dir / $DISSC_CHECKPOINT_DIR

Every file stores inputs (or their seeds), outputs and the library versions.  Afterwards the parity tests that read
these files (tests/test_third_party_pins.py: oracle restatements on the CPU; with --gpu also the HIP path) are run
against the directory just written.  Files land in --out; commit them to turn "parity unpinned" into a pinned row.
"""
import argparse
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FS = 16000
YAAPT_KW = {'frame_length': 20.0, 'frame_space': 5.0, 'nccf_thresh1': 0.25, 'tda_frame_length': 25.0}


def _try(name):
    try:
        return importlib.import_module(name)
    except Exception as e:  # noqa: BLE001  (a broken install counts as absent)
        print(f"  [{name}] not importable: {type(e).__name__}: {e}")
        return None


def _version(mod):
    return str(getattr(mod, "__version__", "unknown"))


# ------------------------------------------------------------------------------------------------------------
# inputs (seeded; stored in the files too, so a reader never has to regenerate them)
# ------------------------------------------------------------------------------------------------------------
def speech_like(f0, seed=0):
    """harmonics up to 3.5 kHz under a two-formant envelope with instantaneous F0 `f0` (per sample) + a little noise"""
    f0 = np.asarray(f0, dtype=np.float64)
    ph = 2 * np.pi * np.cumsum(f0) / FS
    x = np.zeros(len(f0))
    for k in range(1, 60):
        fk = k * f0
        env = 1.0 / (1 + ((fk - 500) / 400) ** 2) + 0.5 / (1 + ((fk - 1500) / 500) ** 2) + 0.05
        x += np.where(fk < 3500, env * np.sin(k * ph + 0.3 * k), 0)
    return 0.1 * x + 1e-3 * np.random.RandomState(seed).standard_normal(len(f0))


def yaapt_inputs(golden_dir):
    from scipy.io import wavfile
    sig = {}
    for name in ("s1_1", "s1_2"):
        p = os.path.join(golden_dir, name + ".wav")
        if os.path.exists(p):
            sr, x = wavfile.read(p)
            assert sr == FS
            sig[name] = (x.astype(np.float32) / 32768.0)
    n = 24000
    sig["flat120"] = speech_like(np.full(n, 120.0), 1).astype(np.float32)
    sig["glide100-250"] = speech_like(np.linspace(100, 250, n), 2).astype(np.float32)
    vu = speech_like(np.full(n, 180.0), 3)
    vu[8000:16000] = 1e-3 * np.random.RandomState(4).standard_normal(8000)  # voiced / unvoiced / voiced
    sig["voiced-unvoiced"] = vu.astype(np.float32)
    return sig


def resample_inputs():
    out = {}
    for sr in (48000, 44100, 22050, 8000):
        n = int(0.25 * sr)
        t = np.arange(n) / sr
        rs = np.random.RandomState(sr)
        out[str(sr)] = sum(0.2 * np.sin(2 * np.pi * f * t + i) for i, f in enumerate((220.0, 1330.0, 3100.0))) \
            + 0.05 * rs.standard_normal(n)
    return out


def trim_input():
    rs = np.random.RandomState(0)
    return np.concatenate([1e-5 * rs.standard_normal(5000), 0.3 * rs.standard_normal(9000), np.zeros(7000)])


# ------------------------------------------------------------------------------------------------------------
# the three pins
# ------------------------------------------------------------------------------------------------------------
def pin_resample(out_dir):
    resampy = _try("resampy")
    if resampy is None:
        return None
    d = {"versions": np.array([f"resampy {_version(resampy)}"])}
    for key, x in resample_inputs().items():
        d[f"in/{key}"] = x
        d[f"out/{key}"] = np.asarray(resampy.resample(x, int(key), FS), dtype=np.float64)  # data/preprocess.py:21-23
    librosa = _try("librosa")
    if librosa is not None:
        x = trim_input()
        y, idx = librosa.effects.trim(x, top_db=20)  # data/preprocess.py:26-27
        d["trim/in"], d["trim/out"], d["trim/index"] = x, np.asarray(y), np.asarray(idx, dtype=np.int64)
        d["versions"] = np.array([f"resampy {_version(resampy)}", f"librosa {_version(librosa)}"])
    path = os.path.join(out_dir, "resample_resampy.npz")
    np.savez_compressed(path, **d)
    return path


def pin_yaapt(out_dir, golden_dir):
    pYAAPT = _try("amfm_decompy.pYAAPT")
    basic = _try("amfm_decompy.basic_tools")
    if pYAAPT is None or basic is None:
        return None
    import amfm_decompy
    d = {"versions": np.array([f"amfm_decompy {_version(amfm_decompy)}"])}
    to_pad = int(YAAPT_KW['frame_length'] / 1000 * FS) // 2
    for name, x in yaapt_inputs(golden_dir).items():
        # reference sr/dataset.py:27-43, one utterance
        y_pad = np.pad(x.astype(np.float64), (to_pad, to_pad), "constant", constant_values=0)
        pitch = pYAAPT.yaapt(basic.SignalObj(y_pad, FS), **YAAPT_KW)
        d[f"in/{name}"] = x.astype(np.float32)
        d[f"f0/{name}"] = np.asarray(pitch.samp_values, dtype=np.float64)
    path = os.path.join(out_dir, "yaapt_amfm.npz")
    np.savez_compressed(path, **d)
    return path


HUBERT_LENGTHS = (400, 719, 4000, 16000, 32000)


def fairseq_hubert(sd, n_layers):
    """fairseq's own HubertModel (base architecture, `n_layers` transformer layers) carrying the state dict `sd`
    (fairseq key layout).  Pre-training-only members (mask_emb, final_proj, label embeddings) keep their init."""
    from fairseq.models.hubert import HubertConfig, HubertModel
    from fairseq.tasks.hubert_pretraining import HubertPretrainingConfig

    class _Dict:  # HubertModel only takes len() of each dictionary (size of the pre-training label set)
        def __len__(self):
            return 504

    cfg = HubertConfig(encoder_layers=n_layers)
    model = HubertModel(cfg, HubertPretrainingConfig(), [_Dict()])
    missing, unexpected = model.load_state_dict(sd, strict=False)
    allowed = ("mask_emb", "final_proj", "label_embs_concat")
    bad = [k for k in missing if not k.startswith(allowed)]
    if bad or unexpected:
        raise RuntimeError(f"synthetic HuBERT weights do not fit fairseq's HubertModel: missing {bad[:5]}, "
                           f"unexpected {list(unexpected)[:5]}")
    return model.eval()


def pin_hubert(out_dir):
    fairseq = _try("fairseq")
    if fairseq is None:
        return None
    import torch
    import synthdata as synth
    sd = synth.synth_hubert_state_dict(6)
    centers = synth.synth_kmeans_centers()
    model = fairseq_hubert(sd, 6)
    from sklearn.cluster import KMeans
    km = KMeans(n_clusters=centers.shape[0], n_init=1)
    km.cluster_centers_ = centers.numpy().astype(np.float32)
    km._n_threads = 1
    km.n_features_in_ = centers.shape[1]
    d = {"versions": np.array([f"fairseq {_version(fairseq)}", f"torch {torch.__version__}"]),
         "weights": np.array(["synthdata.synth_hubert_state_dict(6) / synth_kmeans_centers() (seeded; not stored)"])}
    with torch.no_grad():
        for n in HUBERT_LENGTHS:
            wav = torch.from_numpy(synth.synth_waveform(n, seed=n))[None]
            # what textless' HubertFeatureReader does with a fairseq HuBERT (data/encode.py:21-22,32 of the reference)
            feat, _ = model.extract_features(source=wav, padding_mask=None, mask=False, output_layer=6)
            dense = feat[0].float().numpy()
            d[f"n{n}/dense"] = dense
            d[f"n{n}/units"] = km.predict(dense).astype(np.int64)
    path = os.path.join(out_dir, "hubert_fairseq.npz")
    np.savez_compressed(path, **d)
    return path


def pin_real_encoder(out_dir, golden_dir):
    """The reference's literal encoder call on the speech fixtures with the real checkpoints (whatever textless
    resolves / has cached)."""
    se = _try("textless.data.speech_encoder")
    if se is None:
        return None
    import torch
    from scipy.io import wavfile
    enc = se.SpeechEncoder.by_name(dense_model_name="hubert-base-ls960", quantizer_model_name="kmeans", vocab_size=100,
                                   deduplicate=False)  # data/encode.py:21-22
    d = {"versions": np.array(["textless (unpinned HEAD, reference README.md:31-33)"])}
    for name in ("s1_1", "s1_2"):
        sr, x = wavfile.read(os.path.join(golden_dir, name + ".wav"))
        out = enc(torch.from_numpy(x.astype(np.float32) / 32768.0)[None])  # data/encode.py:32
        d[f"{name}/units"] = out["units"].cpu().numpy().astype(np.int64)
        d[f"{name}/f0"] = out["f0"].cpu().numpy().astype(np.float32)
        d[f"{name}/dense"] = out["dense"].cpu().numpy().astype(np.float32)
    path = os.path.join(out_dir, "hubert_textless_real.npz")
    np.savez_compressed(path, **d)
    return path


# stage -> (libraries, how the reference installs them, the file the stage writes, the tests of tests/test_third_party_pins.py it un-skips,
#           the SURVEY rows that stop being "parity unpinned")
PLAN = {
    "hubert": (["fairseq", "sklearn"],
               "pip install git+https://github.com/facebookresearch/fairseq.git@dd106d9534b22e7db859a6b87ffd7780c38341f8   "
               "(reference README.md:34; scikit-learn: any, comes with textlesslib)",
               "hubert_fairseq.npz", ["test_oracle_hubert_matches_fairseq", "test_hip_hubert_matches_fairseq (-m gpu)"], "a2, a3, a4"),
    "yaapt": (["amfm_decompy"],
              "pip install AMFM-decompy   (unpinned by the reference: a dependency of textlesslib, imported at sr/dataset.py:14-15, "
              "eval.py:11-12; the file records the version used)",
              "yaapt_amfm.npz", ["test_oracle_yaapt_matches_amfm_decompy", "test_hip_yaapt_matches_amfm_decompy (-m gpu)"], "a5, N2"),
    "resample": (["resampy", "librosa"],
                 "pip install resampy librosa   (unpinned by the reference: data/preprocess.py:13,15, sr/dataset.py:226, "
                 "sr/inference.py:20; the file records the versions used)",
                 "resample_resampy.npz", ["test_oracle_resampler_and_trim_match_resampy_librosa", "test_hip_resampler_matches_resampy (-m gpu)"], "N3"),
    "real-encoder": (["textless", "fairseq"],
                     "git clone https://github.com/facebookresearch/textlesslib.git && cd textlesslib && pip install -e .   (reference "
                     "README.md:30-33) + the fairseq pin above; needs the hubert-base-ls960 / km100 downloads (--checkpoint-dir)",
                     "hubert_textless_real.npz", ["test_hip_encoder_matches_textless_with_the_real_checkpoints (-m gpu, $DISSC_CHECKPOINT_DIR)"],
                     "a2-a4 against the reference's literal call (data/encode.py:21-22,32)"),
}


def check(out_dir):
    """--check: nothing is written.  Per stage: is every library importable here (and which version), the exact install line the
    reference names, the file a run would write, whether it already exists, and the tests / SURVEY rows it un-skips."""
    ready = []
    for stage, (libs, pin, fname, tests, rows) in PLAN.items():
        print(f"[{stage}]" + (" (only with --real-encoder)" if stage == "real-encoder" else ""))
        ok = True
        for lib in libs:
            m = _try(lib)
            if m is None:
                ok = False
            else:
                print(f"  {lib} {_version(m)} imports")
        path = os.path.join(out_dir, fname)
        print(f"  reference install: {pin}")
        print(f"  writes: {path}" + ("   (exists: would be overwritten)" if os.path.exists(path) else "   (absent: its tests skip today)"))
        print(f"  un-skips: {'; '.join(tests)}  -> SURVEY rows {rows}")
        print(f"  => {'READY: run without --check' if ok else 'not possible here (library absent): nothing would be written for this stage'}")
        if ok:
            ready.append(stage)
    print(f"ready here: {ready or 'none'}; commit the files written into tests/golden/ to turn the rows into pinned ones")
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--check", action="store_true", help="dry run: per library, what imports here, the reference's pin, the file and tests")
    ap.add_argument("--only", default="hubert,yaapt,resample")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--no-tests", action="store_true", help="write the files only")
    ap.add_argument("--gpu", action="store_true", help="also run the HIP-path tests against the new files (needs the MI355X)")
    ap.add_argument("--real-encoder", action="store_true", help="also pin textless' SpeechEncoder with the real checkpoints")
    ap.add_argument("--checkpoint-dir", default=os.environ.get("DISSC_CHECKPOINT_DIR"))
    a = ap.parse_args(argv)
    if a.check:
        return check(a.out)
    golden = os.path.join(ROOT, "tests", "golden")
    os.makedirs(a.out, exist_ok=True)
    want = [w.strip() for w in a.only.split(",") if w.strip()]
    written, skipped = [], []
    steps = {"resample": lambda: pin_resample(a.out), "yaapt": lambda: pin_yaapt(a.out, golden),
             "hubert": lambda: pin_hubert(a.out)}
    for name in want:
        if name not in steps:
            raise SystemExit(f"--only: unknown stage {name!r} (hubert, yaapt, resample)")
        print(f"[{name}]")
        p = steps[name]()
        (written if p else skipped).append(p or name)
        if p:
            print(f"  wrote {p}")
    if a.real_encoder:
        print("[real encoder]")
        p = pin_real_encoder(a.out, golden)
        (written if p else skipped).append(p or "real-encoder")
    print(f"pinned: {[os.path.basename(p) for p in written]}; skipped (library absent): {skipped}")
    if a.no_tests or not written:
        return 0
    env = dict(os.environ, DISSC_PIN_DIR=os.path.abspath(a.out), DISSC_PIN_NESTED="1")  # NESTED: skip the test that runs this script
    if a.checkpoint_dir:
        env["DISSC_CHECKPOINT_DIR"] = a.checkpoint_dir
    marker = [] if a.gpu else ["-m", "not gpu"]
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_third_party_pins.py"), "-q", "-rs"] + marker
    print("running:", " ".join(cmd))
    return subprocess.call(cmd, env=env, cwd=ROOT)


if __name__ == "__main__":
    sys.exit(main())
