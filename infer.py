#!/usr/bin/env python
"""Speaking-style conversion of HuBERT unit files: rhythm (length) and pitch prediction.

MI355X implementation of the reference's infer.py with the same command line, inputs and
outputs (reference infer.py:175-206): reads a units JSONL, writes ``{out_path}/{basename}``
(reconstruction) and ``{out_path}/{target}_{basename}`` (one file per target speaker) with
lines ``{"units": [...], "f0": [...], "audio": name}``.

Differences in execution only: all (utterance x target) jobs are batched through the HIP
predictors (dissc_amd.predictors.infer_samples) instead of one B=1 call each; JSONL lines are
parsed without eval(); ``-n`` larger than the file is clamped instead of raising IndexError
after the outputs were written (reference infer.py:63 + scripts/convert_eval.py:77).
"""
import argparse
import json
import os
import random

import numpy as np
import torch


def seed_everything(seed):
    """reference utils.py:10-20 (without its tensorflow import)."""
    if seed == -1:
        return
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def prep_stats_tensors(spk_id_dict, f0_param_dict):
    """reference dataset/utils.py:18-26"""
    mean = torch.empty(len(spk_id_dict))
    std = torch.empty(len(spk_id_dict))
    for n, v in spk_id_dict.items():
        mean[v] = f0_param_dict[n]["mean"]
        std[v] = f0_param_dict[n]["std"]
    return mean, std


def _interp_nearest(vals, target_len):
    """reference utils.py:39-45 (scipy interp1d kind='nearest', fill 0)"""
    from scipy.interpolate import interp1d
    cur = len(vals)
    if cur == 1:
        return np.array(target_len * list(vals))
    if target_len == cur:
        return np.array(vals)
    return interp1d(np.linspace(0., 1., cur), vals, bounds_error=False, kind="nearest",
                    fill_value=0)(np.linspace(0., 1., target_len))


def morph_seq_len(units, pitch, t_lens):
    """reference utils.py:47-52: per-run nearest-neighbour resample of the source F0
    (only used for --pred_len without --pred_pitch; host side, off the hot path)."""
    out, i, start = [], 0, 0
    units = list(units)
    for end in range(1, len(units) + 1):
        if end == len(units) or units[end] != units[start]:
            out.append(_interp_nearest(list(pitch[start:end]), int(t_lens[i])))
            i += 1
            start = end
    return np.concatenate(out) if out else np.zeros(0)


def build_models(args, n_speakers, id2mean, id2std):
    from dissc_amd.predictors import LenPredictor, PitchPredictor, PitchPredictorBase
    len_model = pitch_model = None
    if args.pred_len:
        len_model = LenPredictor(n_tokens=args.n_tokens, n_speakers=n_speakers).to(args.device)
        len_model.eval()
        len_model.load_state_dict(torch.load(args.len_model + "best_model.pth", map_location="cpu"))
        len_model.norm_mean, len_model.norm_std = torch.load(args.len_model + "len_norm_stats.pth",
                                                             map_location="cpu")
    if args.pred_pitch:
        cls = PitchPredictorBase if args.f0_model_type == "base" else PitchPredictor
        pitch_model = cls(args.n_tokens, n_speakers, id2pitch_mean=id2mean, id2pitch_std=id2std).to(args.device)
        pitch_model.eval()
        pitch_model.load_state_dict(torch.load(args.f0_model + "best_model.pth", map_location="cpu"))
    return len_model, pitch_model


def run_jobs(jobs, args, len_model, pitch_model, chunk=512):
    """jobs: list of (units, src_f0 or None, spk_id, name, out_file).  Appends one JSON line
    per job to its file, in job order."""
    from dissc_amd.predictors import infer_samples
    for lo in range(0, len(jobs), chunk):
        part = jobs[lo:lo + chunk]
        res = infer_samples([j[0] for j in part], [j[2] for j in part], len_model, pitch_model,
                            norm_pitch=args.norm_pitch, n_tokens=args.n_tokens, device=args.device)
        for (units, f0, lens), (src_units, src_f0, _spk, name, out_file) in zip(res, part):
            if f0 is None:  # --pred_len only: heuristic F0 morphing (reference infer.py:40-41)
                src = [u for u in src_units if u != args.n_tokens]
                f0 = morph_seq_len(src, np.asarray(src_f0, dtype=np.float64), lens).tolist()
            with open(out_file, "a+") as f:
                f.write(json.dumps({"units": units, "f0": f0, "audio": name}) + "\n")


def infer(args):
    from dissc_amd import formats
    spk_id_dict = formats.spk_id_dict_from_list(
        formats.load_pickle(f"{os.path.dirname(args.input_path)}/id_to_spkr.pkl"))
    f0_param_dict = formats.load_pickle(args.f0_path)
    id2mean, id2std = prep_stats_tensors(spk_id_dict, f0_param_dict)
    samples = formats.read_manifest(args.input_path)
    samples = samples[:max(0, min(args.n, len(samples)))]
    base = os.path.basename(args.input_path)
    out_path = f"{args.out_path}/{base}"
    df = None
    if args.sample_df:
        import pandas as pd
        df = pd.read_csv(args.sample_df, index_col=0)
    len_model, pitch_model = build_models(args, len(spk_id_dict), id2mean, id2std)
    targets = None
    if args.vc:
        targets = args.target_speakers or random.sample(list(spk_id_dict.keys()),
                                                        k=min(1, len(spk_id_dict)))
    for p in [out_path] + [f"{args.out_path}/{t}_{base}" for t in (targets or [])]:
        if os.path.exists(p):
            os.remove(p)
    jobs = []
    for s in samples:
        name = s["audio"]
        src = formats.speaker_of(name)
        pitch = np.asarray(s.get("f0", []), dtype=np.float32)
        if args.norm_pitch and len(pitch):
            ii = pitch != 0
            pitch = pitch.copy()
            pitch[ii] = (pitch[ii] - float(id2mean[spk_id_dict[src]])) / float(id2std[spk_id_dict[src]])
        if df is None:
            jobs.append((s["units"], pitch, spk_id_dict[src], name, out_path))
        if targets:
            cur = targets
            if df is not None:
                key = os.path.splitext(name)[0].split("_mic2")[0]
                cur = list(df[df.syn_sample == key].syn_trgt.unique())
            for t in cur:
                jobs.append((s["units"], pitch, spk_id_dict[t], name, f"{args.out_path}/{t}_{base}"))
    run_jobs(jobs, args, len_model, pitch_model)


def infer_wild(args):
    from dissc_amd import formats
    spk_id_dict = formats.spk_id_dict_from_list(formats.load_pickle(args.id_to_spkr))
    id2mean, id2std = prep_stats_tensors(spk_id_dict, formats.load_pickle(args.f0_path))
    len_model, pitch_model = build_models(args, len(spk_id_dict), id2mean, id2std)
    base = os.path.basename(args.input_path)
    for t in args.target_speakers:  # the reference appends to stale files here; we start clean
        p = f"{args.out_path}/{t}_{base}"
        if os.path.exists(p):
            os.remove(p)
    jobs = []
    for s in formats.read_manifest(args.input_path):
        for t in args.target_speakers:
            jobs.append((s["units"], None, spk_id_dict[t], s["audio"], f"{args.out_path}/{t}_{base}"))
    run_jobs(jobs, args, len_model, pitch_model)


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--input_path', default='data/VCTK/hubert100/val.txt', help='Path to txt file of encoded HuBERT data')
    parser.add_argument('-n', default=10, type=int, help='number of samples to perform inference on')
    parser.add_argument('--out_path', default='data/VCTK/pred_hubert', help='Path to save predicted sequence')
    parser.add_argument('--pred_len', action='store_true', help='If true we predict the output length as well')
    parser.add_argument('--pred_pitch', action='store_true', help='If true we predict the output pitch as well')
    parser.add_argument('--len_model', default='checkpoints/vctk/len/', help='Path of len prediction model')
    parser.add_argument('--f0_model', default='checkpoints/vctk/pitch/', help='Path of pitch prediction model & stats')
    parser.add_argument('--f0_model_type', default='new', help='type of model from ["base", "new"]')
    parser.add_argument('--n_tokens', default=100, type=int, help='number of unique HuBERT tokens')
    parser.add_argument('--device', default='cuda:0', help='Device to run on')
    parser.add_argument('--seed', default=42, type=int, help='random seed, use -1 for non-determinism')
    parser.add_argument('--f0_path', default='data/VCTK/hubert100/f0_stats.pkl', help='Pitch normalisation stats pickle')
    parser.add_argument('--vc', action='store_true', help='If true we convert speakers and not only reconstruct')
    parser.add_argument('--norm_pitch', action='store_false', help='If true we output a per-speaker normalised pitch')
    parser.add_argument('--target_speakers', nargs='+', default=None, help='Target speakers for VC')
    parser.add_argument('--sample_df', default=None, help='Path for specific conversions for each sample')
    parser.add_argument('--wild_sample', action='store_true', help='convert a new sample from an unknown speaker')
    parser.add_argument('--id_to_spkr', default=None, help='Path of id to spkr pickle, used for wild samples only')
    args = parser.parse_args(argv)

    assert args.pred_len | args.pred_pitch, "Inference must at least convert pitch or rhythm (or both)"
    from dissc_amd.harness import limit_host_threads
    limit_host_threads()  # torch's host pool: a few threads, not one per logical CPU (start-up cost, harness.py)
    assert (args.wild_sample & args.pred_len & args.pred_pitch) | (not args.wild_sample), \
        "If we use an unknown speaker we must convert both pitch and rhythm"
    seed_everything(args.seed)
    os.makedirs(args.out_path, exist_ok=True)
    p = f"{args.out_path}/{os.path.basename(args.input_path)}"
    if os.path.exists(p):
        os.remove(p)
    if args.wild_sample:
        infer_wild(args)
    else:
        infer(args)


if __name__ == '__main__':
    main()
