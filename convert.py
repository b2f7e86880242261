#!/usr/bin/env python
"""wav directory -> converted wav directory: the whole DISSC conversion (HuBERT units -> rhythm +
pitch prediction -> HiFi-GAN resynthesis) in one process per GPU, without the JSONL files in between.

Equivalent to the reference's three-script chain for an unseen source speaker (reference README.md
73-85, scripts/convert_eval.py:77,93)::

    python3 data/encode.py --base_dir W --out_file E/enc.txt
    python3 infer.py --input_path E/enc.txt --pred_len --pred_pitch --vc --wild_sample \\
            --id_to_spkr I --target_speakers T... --out_path P
    python3 sr/inference.py --input_code_file P/T_enc.txt --vc --unseen_speaker --id_to_spkr I \\
            --target-speakers T --output_dir O          (once per target)

and writes the same files, ``O/{stem}_{target id}_gen.wav`` (float32, 16 kHz, peak-normalised),
sample for sample (tests/test_gpu_pipeline.py).  The file-based scripts remain for format parity.

Multi-GPU (BASELINE.json configs[3], [4]): launch with ``python -m torch.distributed.run
--nproc-per-node N convert.py ...``; utterances are LPT-sharded over the ranks by length, every
rank runs encode -> predict -> resynthesise for its share, and rank 0 receives every waveform
through ONE all-gather per round (dissc_amd/pipeline.py, dissc_amd/harness.py) and writes the files; a run is a
single round unless a rank's share exceeds --round_seconds.  ``DISSC_FORCE_DIST=1`` runs the collectives on RCCL
with a single rank too.
"""
import argparse
import json
import os
import sys
import wave

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "data")):
    if p not in sys.path:
        sys.path.insert(0, p)


def wav_frames(path):
    """sample count from the header only (every rank needs all lengths, but loads only its share)"""
    try:
        with wave.open(path, "rb") as w:
            return w.getnframes()
    except (wave.Error, EOFError):
        from encode import load_wav
        return len(load_wav(path)[0])


def build_parser():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--base_dir", required=True, help="directory of 16 kHz wavs (data/encode.py --base_dir)")
    ap.add_argument("--output_dir", required=True, help="where {stem}_{id}_gen.wav are written (sr/inference.py)")
    ap.add_argument("--hubert_dir", default=None, help="HuBERT / k-means checkpoint directory (data/encode.py --checkpoint_dir)")
    ap.add_argument("--model_name", default="hubert-base-ls960")
    ap.add_argument("--quantizer_name", default="kmeans")
    ap.add_argument("--vocab_size", default=100, type=int)
    ap.add_argument("--len_model", default="checkpoints/vctk/len/", help="infer.py --len_model")
    ap.add_argument("--f0_model", default="checkpoints/vctk/pitch/", help="infer.py --f0_model")
    ap.add_argument("--f0_model_type", default="new", help='"base" or "new" (infer.py --f0_model_type)')
    ap.add_argument("--n_tokens", default=100, type=int)
    ap.add_argument("--checkpoint_file", default="checkpoints/vctk_hubert/", help="vocoder dir or g_* file (sr/inference.py)")
    ap.add_argument("--id_to_spkr", required=True, help="pickled speaker list (index = id)")
    ap.add_argument("--target_speakers", nargs="+", required=True)
    ap.add_argument("--no_pred_len", action="store_true", help="keep the source rhythm (infer.py without --pred_len)")
    ap.add_argument("--round_seconds", default=16384.0, type=float,
                    help="input audio x targets per rank and round: each round is gathered, written and freed.  TWO "
                         "collectives per round: a 16-byte all_reduce(MAX) of the exchange-buffer geometry (predicted "
                         "durations size the outputs, so ranks cannot derive it from the file list) and ONE "
                         "all_gather_into_tensor of the packed waveforms; the default holds 1 GiB of waveforms per rank.  "
                         "Rounds are delivered (device-to-host copy, file writes) by a worker thread while the next one "
                         "computes; env DISSC_WRITERS=rank0|all picks who writes at N > 1 (default all: every rank "
                         "writes the conversions it produced)")
    return ap


def main(argv=None):
    a = build_parser().parse_args(argv)
    from dissc_amd import harness
    rank, local_rank, world, dist = harness.init_distributed(29513)
    device = torch.device("cuda", local_rank)

    from encode import load_wav
    from sr.inference import scan_checkpoint
    from dissc_amd import AttrDict, CodeGenerator, formats
    from dissc_amd.hubert import SpeechEncoder
    from dissc_amd.pipeline import Converter
    from dissc_amd.predictors import LenPredictor, PitchPredictor, PitchPredictorBase

    id_to_spkr = formats.load_pickle(a.id_to_spkr)
    spk_id = formats.spk_id_dict_from_list(id_to_spkr)
    targets = [spk_id[t] for t in a.target_speakers]

    enc = SpeechEncoder.by_name(a.model_name, a.quantizer_name, a.vocab_size, checkpoint_dir=a.hubert_dir).to(device)
    len_model = None
    if not a.no_pred_len:
        len_model = LenPredictor(n_tokens=a.n_tokens, n_speakers=len(spk_id)).to(device)
        len_model.eval()
        len_model.load_state_dict(torch.load(a.len_model + "best_model.pth", map_location="cpu"))
        len_model.norm_mean, len_model.norm_std = torch.load(a.len_model + "len_norm_stats.pth", map_location="cpu")
    cls = PitchPredictorBase if a.f0_model_type == "base" else PitchPredictor
    pitch_model = cls(a.n_tokens, len(spk_id)).to(device)
    pitch_model.eval()
    pitch_model.load_state_dict(torch.load(a.f0_model + "best_model.pth", map_location="cpu"))

    if os.path.isdir(a.checkpoint_file):
        config_file, cp_g = os.path.join(a.checkpoint_file, "config.json"), scan_checkpoint(a.checkpoint_file, "g_")
    else:
        config_file, cp_g = os.path.join(os.path.split(a.checkpoint_file)[0], "config.json"), a.checkpoint_file
    with open(config_file) as f:
        h = AttrDict(json.loads(f.read()))
    generator = CodeGenerator(h).to(device)
    generator.load_state_dict(torch.load(cp_g, map_location="cpu")["generator"])
    generator.eval()
    generator.remove_weight_norm()

    files = sorted(f for f in os.listdir(a.base_dir) if f.lower().endswith(".wav"))
    n_samples = [wav_frames(os.path.join(a.base_dir, f)) for f in files]
    keep = [i for i, n in enumerate(n_samples) if n >= 400]
    for i in sorted(set(range(len(files))) - set(keep)):
        if rank == 0:
            print(f"Problem encoding sample {files[i]}: shorter than one HuBERT frame")
    files, n_samples = [files[i] for i in keep], [n_samples[i] for i in keep]

    def load(i):
        x, sr = load_wav(os.path.join(a.base_dir, files[i]))
        if sr != 16000:
            raise ValueError(f"{files[i]}: sample rate {sr}, expected 16000 (run data/preprocess.py first)")
        return x

    # the vocoder data set's F0 normalisation by the SOURCE speaker's statistics (reference sr/dataset.py:255-267,
    # sr/inference.py build_jobs here): the shipped configs set f0_normalize with f0_stats, and the chain applies it
    # to whatever F0 the manifest holds -- so does this entry
    f0_stats = None
    if h.get("f0_normalize", False) and h.get("f0_stats", None):
        import pickle
        with open(h["f0_stats"], "rb") as f:
            st = pickle.load(f)
        pairs = []
        for fn in files:
            e = st.get(formats.parse_speaker(os.path.join(a.base_dir, fn), h.get("multispkr", None) or "_"), None)
            pairs.append((e["mean"], e["std"]) if e is not None else (st["f0_mean"], st["f0_std"]))
        f0_stats = (np.array([p[0] for p in pairs], np.float64), np.array([p[1] for p in pairs], np.float64))

    conv = Converter(enc.model, len_model, pitch_model, generator, norm_pitch=True, n_tokens=a.n_tokens,
                     f0_median=bool(h.get("f0_median", False)))
    if rank == 0:
        os.makedirs(a.output_dir, exist_ok=True)

    def write(waves):  # once per round, on the harness's delivery thread while the next round is computed
        for (i, t), w in sorted(waves.items()):
            formats.write_wav(os.path.join(a.output_dir, f"{os.path.splitext(files[i])[0]}_{t}_gen.wav"),
                              h.sampling_rate, w)

    # DISSC_WRITERS=all (default for N > 1): every rank writes the conversions it produced (it drains its own packed
    # buffer; the round's all-gather then carries the row tables only); rank0: rank 0 receives and writes every file
    own = world > 1 and os.environ.get("DISSC_WRITERS", "all") != "rank0"
    n = conv.run_sharded(n_samples, load, targets, rank, world, dist, f0_stats=f0_stats, sink=write,
                         round_floats=int(a.round_seconds * 16000), own_rows=own)
    if own or rank == 0:
        print(f"rank {rank}: {n} waveforms written to {a.output_dir}" if own else f"{n} waveforms written to {a.output_dir}")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
