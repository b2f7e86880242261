#!/usr/bin/env python
"""Unit/F0 JSONL -> waveforms with the HiFi-GAN generator on MI355X.

Same command line, inputs and outputs as the reference's sr/inference.py
(reference sr/inference.py:259-359): reads ``<checkpoint dir>/config.json`` + the latest
``g_########`` checkpoint, parses the manifest, and writes float32 16 kHz WAVs
``{stem}_gen.wav`` (resynthesis), ``{stem}_{spk_id}_gen.wav`` (one per target speaker) and
``{stem}_gt.wav`` (when the ground-truth wav exists under --data_path).

Execution differs: instead of ``Pool(8)`` B=1 workers, launch one process per GPU
(``python -m torch.distributed.run --nproc-per-node N sr/inference.py ...``; a plain
``python sr/inference.py`` is the 1-GPU case).  Jobs are LPT-sharded over ranks, batched through
the HIP generator, post-processed on the GPU and exchanged with ONE RCCL all-gather per round (a run
is 1-4 rounds; a round is delivered -- device-to-host copy, file writes -- by a worker thread while the
next one computes).  Like the reference's pool workers every rank writes the files of the jobs it
decoded (DISSC_WRITERS=all, the default for N > 1; DISSC_WRITERS=rank0: rank 0 receives and writes
everything).  The ground-truth mel the reference computes and throws
away (sr/dataset.py:269-271) is not computed.
"""
import argparse
import glob
import json
import os
import pickle
import random
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MAX_WAV_VALUE = 32768.0


def scan_checkpoint(cp_dir, prefix):
    """latest checkpoint by sorted glob (reference sr/inference.py:59-64)"""
    cp_list = glob.glob(os.path.join(cp_dir, prefix + '*'))
    return sorted(cp_list)[-1] if cp_list else ''


def peak_normalize(x):
    """librosa.util.normalize for a 1-D signal (reference sr/inference.py:206,250,255)"""
    x = np.asarray(x, dtype=np.float32)
    peak = np.max(np.abs(x)) if x.size else 0.0
    return x / peak if peak >= np.finfo(np.float32).tiny else x


def load_gt(path, code_len, pad=None, sampling_rate=16000, code_hop_size=None):
    """Ground-truth audio exactly as CodeDataset.__getitem__ prepares it (reference
    sr/dataset.py:221-264,199-219): int16 -> /32768 -> peak*0.95 -> [eval_mode False only: trim to
    min(len//code_hop_size, code_len) hops] -> trim to a whole number of hops.  None when the wav is
    missing (the reference would crash).  A wav at another rate is resampled like the reference does
    (sr/dataset.py:225-227; resampy restated, parity unpinned)."""
    if not os.path.isfile(path):
        return None
    from scipy.io import wavfile  # (only the ground-truth copies read wavs: not on the start-up path of a conversion)
    sr, audio = wavfile.read(path)
    if audio.ndim > 1:
        audio = audio[:, 0]
    if audio.dtype != np.int16:  # the reference reads with dtype='int16' (sr/dataset.py:97-105)
        audio = (np.clip(audio, -1, 1) * 32767).astype(np.int16) if audio.dtype.kind == 'f' else audio.astype(np.int16)
    if sr != sampling_rate:
        # reference sr/dataset.py:225-227 (resampy.resample of the int16-valued samples -> float64): same kernel
        # as data/preprocess.py (dissc_amd.audio.resample; parity with resampy unpinned)
        from dissc_amd import audio as _audio
        audio = _audio.resample(audio.astype(np.float64), sr, sampling_rate)
    if pad:
        audio = np.pad(audio, (0, pad - (audio.shape[-1] % pad)), "constant")
    audio = audio / MAX_WAV_VALUE
    peak = np.max(np.abs(audio))
    audio = (audio / peak if peak >= np.finfo(audio.dtype).tiny else audio) * 0.95
    audio = audio.astype(np.float32)
    if code_hop_size:  # eval_mode False (reference sr/dataset.py:246-253)
        code_len = min(audio.shape[0] // code_hop_size, code_len)
        audio = audio[:code_len * code_hop_size]
    n = audio.shape[0]
    if code_len > 0 and n >= code_len:
        hop = n // code_len
        audio = audio[:(n // hop) * hop]
    return audio


def build_jobs(a, h, samples, id_to_spkr, f0_stats_cfg, target_f0_stats):
    """Expand the manifest into generator jobs in the reference's output order."""
    from dissc_amd import formats
    spkr_to_id = {k: v for v, k in enumerate(id_to_spkr)}
    df = None
    if a.sample_df:
        import pandas as pd
        df = pd.read_csv(a.sample_df, index_col=0)
        if a.target_speakers:
            df = df[df.syn_trgt.isin(a.target_speakers)]
    spkrs = None
    if h.get('multispkr', None):
        if a.target_speakers is not None:
            spkrs = [spkr_to_id[s] for s in a.target_speakers]
        else:
            spkrs = random.sample(range(len(id_to_spkr)), k=min(5, len(id_to_spkr)))
    base_path = h.get('test_base_path', '') if a.data_path is None else a.data_path
    # Which items run when -n is smaller than the manifest: the reference walks a shuffled
    # index (random.seed(1234) from CodeDataset.__init__, sr/dataset.py:156; shuffle at
    # sr/inference.py:351-352) and stops after n+1 results (:356-357); --debug walks in order
    # and stops after n+2 (:347-348).
    order = list(range(len(samples)))
    if a.debug:
        keep = order if a.n == -1 else order[:a.n + 2]
    else:
        random.seed(1234)
        random.shuffle(order)
        keep = order if a.n == -1 else order[:a.n + 1]
    keep = sorted(keep)
    jobs, items = [], []
    for idx in keep:
        s = samples[idx]
        audio_path = Path(str(base_path) + '/' + s['audio'].split('/')[-1])
        if a.parts:
            stem = '_'.join(audio_path.parts[-3:])[:-4]
        else:
            stem = audio_path.stem
        code = np.asarray(s['units'], dtype=np.int64)
        f0 = np.asarray(s.get('f0', np.zeros(len(code))), dtype=np.float32).copy()
        if len(f0) != len(code):
            raise ValueError(f"{s['audio']}: {len(code)} units but {len(f0)} f0 values")
        if not a.eval_mode:
            # --eval_mode is store_false: when given, code and pitch are clipped to the ground-truth
            # audio (reference sr/dataset.py:243-251); needs the wav like the reference does
            hop = int(h.code_hop_size)
            gt = load_gt(str(audio_path), len(code), a.pad, h.sampling_rate, hop)
            if gt is None:
                raise FileNotFoundError(f"--eval_mode needs the ground-truth audio {audio_path}")
            code_len = min(len(gt) // hop, len(code))
            code, f0 = code[:code_len], f0[:code_len]
        # speaker rule of the vocoder config (reference sr/dataset.py:132-147); configs without `multispkr` have no
        # speaker embedding, the name then only selects F0 statistics (the '_' rule, like every shipped config)
        src_name = formats.parse_speaker(audio_path, h.get('multispkr', None) or '_')
        if h.get('f0_normalize', False) and f0_stats_cfg is not None:
            st = f0_stats_cfg.get(src_name, None)
            mean, std = (st['mean'], st['std']) if st is not None else (f0_stats_cfg['f0_mean'], f0_stats_cfg['f0_std'])
            ii = f0 != 0
            if h.get('f0_median', False) and ii.any():
                f0[~ii] = (np.median(f0[ii]) - mean) / std
            f0[ii] = (f0[ii] - mean) / std
        emits_src = a.sample_df is None and not a.unseen_speaker
        if a.unseen_speaker or not h.get('multispkr', None):
            src_id = 0  # reference sr/dataset.py:292-293 (unseen) / no speaker embedding at all
        else:  # the reference's data set looks the source speaker up for every item, converted or not
            src_id = formats.speaker_id(src_name, spkr_to_id)  # KeyError like sr/dataset.py:319-322
        items.append((stem, audio_path, len(code)))
        if emits_src:
            jobs.append(dict(code=code, f0=f0, spkr=src_id, out=f"{stem}_gen.wav"))
        if h.get('multispkr', None) and a.vc:
            local = spkrs if a.target_speakers is not None else \
                random.sample(range(len(id_to_spkr)), k=min(5, len(id_to_spkr)))
            if df is not None:
                local = [spkr_to_id[i] for i in df[df.syn_sample == stem.split('_mic2')[0]].syn_trgt.unique()]
            for k in local:
                f0k = f0
                if target_f0_stats is not None and h.get('f0', None) is not None and not h.get('f0_normalize', False):
                    # re-normalise voiced F0 to the target speaker (reference sr/inference.py:221-236)
                    f0k = f0.copy()
                    ii = f0k != 0
                    if ii.any():
                        v = torch.from_numpy(f0k[ii])
                        mean_, std_ = float(v.mean()), float(v.std())
                        st = target_f0_stats.get(k, None)
                        nm, ns = (st['f0_mean'], st['f0_std']) if st is not None else \
                            (target_f0_stats['f0_mean'], target_f0_stats['f0_std'])
                        f0k[ii] = (f0k[ii] - mean_) / std_ * float(ns) + float(nm)
                jobs.append(dict(code=code, f0=f0k, spkr=int(k), out=f"{stem}_{k}_gen.wav"))
    return jobs, items


class _Phases:
    """DISSC_CLI_TIMING=1: wall-clock marks since PROCESS START (tools/cli_wall.py), printed as one JSON line at exit."""

    def __init__(self):
        self.on = os.environ.get('DISSC_CLI_TIMING') == '1'
        self.marks = []
        if self.on:
            import time

            import psutil
            self.t0 = psutil.Process().create_time()
            self.now = time.time
            self.mark('imports')

    def mark(self, name, sync=None):
        if self.on:
            if sync is not None and torch.cuda.is_available():
                torch.cuda.synchronize(sync)
            self.marks.append((name, self.now() - self.t0))

    def dump(self, rank, extra):
        if self.on:
            prev, out = 0.0, {}
            for name, t in self.marks:
                out[name + '_s'] = round(t - prev, 4)
                prev = t
            print('CLI_TIMING ' + json.dumps(dict(out, total_s=round(prev, 4), rank=rank, **extra)), flush=True)


def main(argv=None):
    ph = _Phases()
    print('Initializing Inference Process..')
    parser = argparse.ArgumentParser()
    parser.add_argument('--code_file', default=None)
    parser.add_argument('--input_code_file', default='data/wild/pred_hubert/p239_encoded.txt')
    parser.add_argument('--data_path', default=None, help='Base path for the wavs to override config')
    parser.add_argument('--output_dir', default='debug')
    parser.add_argument('--checkpoint_file', default='checkpoints/vctk_hubert/')
    parser.add_argument('--f0-stats', type=Path)
    parser.add_argument('--vc', action='store_true')
    parser.add_argument('--target-speakers', default=None, nargs='+',
                        help='target speakers, if None, 5 random speakers are chosen')
    parser.add_argument('--pad', default=None, type=int)
    parser.add_argument('--debug', action='store_true')
    parser.add_argument('--eval_mode', action='store_false',
                        help='If true the samples are generated and not clipped to a given length')
    parser.add_argument('--parts', action='store_true')
    parser.add_argument('--unseen-f0', type=Path)
    parser.add_argument('--unseen_speaker', action='store_true',
                        help='the input for conversion is an unseen speaker')
    parser.add_argument('--id_to_spkr', default=None, type=Path,
                        help='Path for id_to_spkr pickle. Used for unseen speakers')
    parser.add_argument('--sample_df', default=None, type=Path)
    parser.add_argument('-n', type=int, default=2508)
    a = parser.parse_args(argv)

    seed = 52
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)

    from dissc_amd import harness
    rank, local_rank, world, dist = harness.init_distributed(29512)
    device = torch.device('cuda', local_rank)
    ph.mark('process_group')
    # The HIP context (+ the first allocation) takes ~0.3 s and needs nothing from the host work below -- config, manifest,
    # checkpoint load --, so it is created on a helper thread meanwhile and joined before the first tensor goes to the device.
    import threading
    hip_up = threading.Thread(target=lambda: torch.zeros(1, device=device), name='dissc-hip-init', daemon=True)
    hip_up.start()

    if os.path.isdir(a.checkpoint_file):
        config_file = os.path.join(a.checkpoint_file, 'config.json')
        cp_g = scan_checkpoint(a.checkpoint_file, 'g_')
    else:
        config_file = os.path.join(os.path.split(a.checkpoint_file)[0], 'config.json')
        cp_g = a.checkpoint_file
    from dissc_amd import AttrDict, CodeGenerator, formats, harness
    from dissc_amd.generator import wav_postprocess_
    with open(config_file) as f:
        h = AttrDict(json.loads(f.read()))
    if not os.path.isfile(cp_g):
        print(f"Didn't find checkpoints for {cp_g}")
        return

    if a.code_file is not None:
        # "name|c c c ..." lines: units only (reference sr/inference.py:122-129)
        samples = []
        for line in open(a.code_file):
            name, codes = line.strip().split('|')[:2]
            samples.append({'audio': name, 'units': [int(v) for v in codes.split(' ')]})
        id_to_spkr = []
    else:
        samples = [s for s in formats.read_manifest(a.input_code_file) if 'units' in s]
        if a.unseen_speaker:
            id_to_spkr = formats.load_pickle(a.id_to_spkr)
        else:
            id_to_spkr = formats.load_pickle(f'{os.path.dirname(h.input_training_file)}/id_to_spkr.pkl')
    f0_stats_cfg = None
    if h.get('f0_normalize', False) and h.get('f0_stats', None):
        with open(h['f0_stats'], 'rb') as f:
            f0_stats_cfg = pickle.load(f)
    target_f0_stats = torch.load(a.f0_stats) if a.f0_stats else None

    print("Loading '{}'".format(cp_g))
    state = torch.load(cp_g, map_location='cpu')
    print("Complete.")
    ph.mark('manifest_and_checkpoint_load_host')
    hip_up.join()
    ph.mark('hip_init_exposed', device)  # what is left of the context creation after the host work above
    generator = CodeGenerator(h).to(device)
    generator.load_state_dict(state['generator'])
    generator.eval()
    generator.remove_weight_norm()
    generator.prepare()
    ph.mark('fold_and_pack_weights', device)

    os.makedirs(a.output_dir, exist_ok=True)
    jobs, items = build_jobs(a, h, samples, id_to_spkr, f0_stats_cfg, target_f0_stats)
    ph.mark('build_jobs')

    def write(waves):  # once per round, on the harness's delivery thread while the next round is computed
        for j, w in sorted(waves.items()):
            formats.write_wav(os.path.join(a.output_dir, jobs[j]['out']), h.sampling_rate, w)

    # Who writes (DISSC_WRITERS): "all" (default for N > 1, like the reference's pool workers, which each write their
    # own outputs: sr/inference.py:205-207,249-251 there) -- every rank drains the rows it decoded itself (the round's
    # all-gather then carries the row tables only), so device-to-host copies and file writes spread over the ranks; "rank0" -- rank 0 receives and writes
    # every file.  The files are byte-identical either way.
    own = world > 1 and os.environ.get('DISSC_WRITERS', 'all') != 'rank0'
    run_stats = {} if ph.on else None
    harness.run_resynthesis(generator, jobs, rank, world, device, dist, postprocess=wav_postprocess_, sink=write,
                            own_rows=own, stats=run_stats)
    ph.mark('resynthesis_and_writes', device)
    if a.sample_df is None:  # ground-truth copies: no GPU work, the items are dealt round-robin to the writers
        for stem, audio_path, code_len in (items[rank::world] if own else items if rank == 0 else []):
            gt = load_gt(str(audio_path), code_len, a.pad, h.sampling_rate,
                         None if a.eval_mode else int(h.code_hop_size))
            if gt is not None:
                formats.write_wav(os.path.join(a.output_dir, stem + '_gt.wav'), h.sampling_rate,
                                  peak_normalize(gt))
    ph.mark('gt_copies')
    if rank == 0:
        print(f'{len(jobs)} waveforms written to {a.output_dir}')
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ph.mark('teardown')
    ph.dump(rank, {'jobs': len(jobs), 'world': world,
                   'run': {k: round(float(v), 4) for k, v in (run_stats or {}).items() if isinstance(v, (int, float))}})


if __name__ == '__main__':
    main()
