#!/usr/bin/env python
"""Train the pitch predictor on the MI355X.  Same command line, inputs and outputs as the reference's
train_f0_predictor.py (reference train_f0_predictor.py:14-124): reads ``{data_path}/train.txt``, ``val.txt``,
``id_to_spkr.pkl`` and the per-speaker F0 statistics pickle, writes ``{out_path}/pitch/best_model.pth`` (lowest
validation MAE).  Every optimisation step is one call into libdissc_hip.so (dissc_amd/train.py); validation runs the
inference kernels.  Metrics go to ``{out_path}/pitch/log.jsonl``."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def train(data_path, f0_path, device='cuda:0', args=None):
    from infer import prep_stats_tensors
    from dissc_amd import formats
    from dissc_amd.predictors import PitchPredictor, PitchPredictorBase
    from dissc_amd.train import Trainer, batches, init_state_dict, load_pitch_dataset, write_log
    pad = -100
    # --seed -1 = non-deterministic, as documented (reference train_*_predictor.py --seed help): draw one
    run_seed = args.seed if args.seed >= 0 else torch.seed() % (1 << 31)
    out_path = args.out_path + '/pitch'
    f0_param_dict = formats.load_pickle(f0_path)
    spk_id_dict = formats.spk_id_dict_from_list(formats.load_pickle(f'{args.data_path}/id_to_spkr.pkl'))
    id2mean, id2std = prep_stats_tensors(spk_id_dict, f0_param_dict)
    tr = load_pitch_dataset(f'{data_path}/train.txt', spk_id_dict, f0_param_dict, args.n_tokens, pad)
    va = load_pitch_dataset(f'{data_path}/val.txt', spk_id_dict, f0_param_dict, args.n_tokens, pad)
    kind = 'base' if args.model_type == 'base' else 'new'
    trainer = Trainer(kind, init_state_dict(kind, args.n_tokens, len(spk_id_dict)), args.learning_rate,
                      stats=(id2mean, id2std), seed=run_seed).to(device)
    gen = torch.Generator().manual_seed(run_seed)
    log = out_path + '/log.jsonl'
    if os.path.exists(log):
        os.remove(log)
    best_mae = float('inf')
    cls = PitchPredictorBase if kind == 'base' else PitchPredictor
    for epoch in range(args.n_epochs):
        print(f'\nEpoch: {epoch}')
        tot, n_samples = None, 0
        nb = (len(tr[0]) + args.batch_size - 1) // args.batch_size
        for i, idx in enumerate(batches(len(tr[0]), args.batch_size, True, gen)):
            seqs, gts, spk = tr[0][idx], tr[1][idx], tr[2][idx]
            loss = trainer.step(seqs, spk, gts, pad_value=pad)
            cur = int((gts != pad).sum())
            n_samples += cur
            tot = loss if tot is None else tot + loss
            print(f'\r finished: {100 * i / nb:.2f}%, train loss: {float(loss) / max(cur, 1):.5f}', end='')
        print()
        sd = trainer.state_dict()
        model = cls(args.n_tokens, len(spk_id_dict), id2pitch_mean=id2mean, id2pitch_std=id2std).to(device)
        model.load_state_dict(sd)
        results = {}
        for split, (vals, gts_all, spk_all, _) in (('train', tr), ('val', va)):
            mae = mse = 0.0
            n = 0
            for idx in batches(len(vals), args.batch_size, False):
                gts, spk = gts_all[idx], spk_all[idx].long()
                freqs = model.infer_freq(vals[idx].long(), spk, False).cpu()  # Hz, 0 = unvoiced (calc_freq)
                mask = gts != pad
                want = (id2mean[spk] + id2std[spk] * gts) * (gts != 0)    # PitchMAE / PitchMSE, reference loss/pitch_loss.py
                mae += float((mask * (freqs - want).abs()).sum())
                mse += float((mask * (freqs - want) ** 2).sum())
                n += int(mask.sum())
            results[split] = {'MAE': mae / max(n, 1), 'MSE': mse / max(n, 1)}
            results[split + '_total_mae'] = mae
        results['train']['loss'] = float(tot) / max(n_samples, 1)
        if results['val_total_mae'] < best_mae:
            torch.save(sd, out_path + '/best_model.pth')
            best_mae = results['val_total_mae']
        write_log(log, 'train', epoch, results['train'])
        write_log(log, 'val', epoch, results['val'])


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--out_path', default='checkpoints/vctk', help='Path to save model and logs')
    parser.add_argument('--data_path', default='data/VCTK/hubert100/', help='Path to sequence data')
    parser.add_argument('--f0_path', default='data/VCTK/hubert100/f0_stats.pkl', help='Pitch normalisation stats pickle')
    parser.add_argument('--model_type', default='base', help='type of model from ["base", "new"]. New has PE and few other modifications')
    parser.add_argument('--n_tokens', default=100, type=int, help='number of unique HuBERT tokens to use (which represent how many clusters were used)')
    parser.add_argument('--device', default='cuda:0', help='Device to run on')
    parser.add_argument('--seed', default=42, type=int, help='random seed, use -1 for non-determinism')
    parser.add_argument('--batch_size', default=32, type=int, help='batch size for train and inference')
    parser.add_argument('--learning_rate', default=3e-4, type=float, help='initial learning rate of the Adam optimiser')
    parser.add_argument('--n_epochs', default=30, type=int, help='number of training epochs')
    args = parser.parse_args(argv)
    from infer import seed_everything
    seed_everything(args.seed)
    os.makedirs(args.out_path, exist_ok=True)
    os.makedirs(args.out_path + '/pitch', exist_ok=True)
    train(args.data_path, args.f0_path, args.device, args)


if __name__ == '__main__':
    main()
