#!/usr/bin/env python
"""Train the rhythm (length) predictor on the MI355X.  Same command line, inputs and outputs as the reference's
train_len_predictor.py (reference train_len_predictor.py:13-128): reads ``{data_path}/train.txt``, ``val.txt`` and
``id_to_spkr.pkl``, writes ``{out_path}/len/best_model.pth`` (the module's state_dict, lowest validation MSE) and
``len_norm_stats.pth``.  Every optimisation step is one call into libdissc_hip.so (dissc_amd/train.py); validation
runs the inference kernels.  Metrics go to ``{out_path}/len/log.jsonl`` (the reference writes TF summaries;
tensorflow is not available here)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def len_metrics(preds, lens, pad=-1):
    """LenSumLoss / LenMSELoss / LenMAELoss / LenExactAccuracy / LenOneOffAccuracy (reference loss/len_loss.py) as
    plain tensor arithmetic on whatever device the predictions live on"""
    import torch.nn.functional as F
    mask = lens != pad
    diff = preds - lens
    diff4 = (F.avg_pool2d(diff.unsqueeze(0), (1, 4)) * 4) ** 2
    mask4 = ~F.max_pool2d((lens == pad).unsqueeze(0).float(), (1, 4)).bool()
    mse = (mask * diff ** 2).sum()
    rounded = torch.round(torch.clamp(preds, min=1)).int()
    return {"Loss": mse + 0.5 * (mask4 * diff4).sum(), "MSE": mse, "MAE": (mask * diff.abs()).sum(),
            "Accuracy": (mask * (rounded == lens)).sum(), "Accuracy_1": (mask * ((rounded - lens).abs() <= 1)).sum()}


def train(data_path, device='cuda:0', args=None):
    from infer import seed_everything  # noqa: F401  (seeded in main)
    from dissc_amd import formats
    from dissc_amd.predictors import LenPredictor
    from dissc_amd.train import Trainer, batches, init_state_dict, load_len_dataset, write_log
    pad = -1
    # --seed -1 = non-deterministic, as documented (reference train_*_predictor.py --seed help): draw one
    run_seed = args.seed if args.seed >= 0 else torch.seed() % (1 << 31)
    out_path = args.out_path + '/len'
    spk_id_dict = formats.spk_id_dict_from_list(formats.load_pickle(f'{args.data_path}/id_to_spkr.pkl'))
    tr_vals, tr_lens, tr_spk, _ = load_len_dataset(f'{data_path}/train.txt', spk_id_dict, args.n_tokens, pad)
    va_vals, va_lens, va_spk, _ = load_len_dataset(f'{data_path}/val.txt', spk_id_dict, args.n_tokens, pad)
    valid = tr_lens[tr_lens != pad]
    norm_mean, norm_std = valid.mean(), valid.std()
    torch.save((norm_mean, norm_std), out_path + '/len_norm_stats.pth')
    trainer = Trainer('len', init_state_dict('len', args.n_tokens, len(spk_id_dict)), args.learning_rate,
                      norm=(norm_mean, norm_std), seed=run_seed).to(device)
    gen = torch.Generator().manual_seed(run_seed)
    log = out_path + '/log.jsonl'
    if os.path.exists(log):
        os.remove(log)
    best_mse = float('inf')
    for epoch in range(args.n_epochs):
        print(f'Epoch: {epoch}')
        tot, n_samples = None, 0
        nb = (len(tr_vals) + args.batch_size - 1) // args.batch_size
        for i, idx in enumerate(batches(len(tr_vals), args.batch_size, True, gen)):
            seqs, lens, spk = tr_vals[idx], tr_lens[idx], tr_spk[idx]
            loss = trainer.step(seqs, spk, lens, pad_value=pad)
            cur = int((seqs != args.n_tokens).sum())
            n_samples += cur
            tot = loss if tot is None else tot + loss
            print(f'\r finished: {100 * i / nb:.2f}%, train loss: {float(loss) / max(cur, 1):.5f}', end='')
        print()
        # metrics of the epoch with the inference kernels (eval-mode BatchNorm) on the updated weights
        sd = trainer.state_dict()
        model = LenPredictor(args.n_tokens, len(spk_id_dict)).to(device)
        model.load_state_dict(sd)
        model.norm_mean, model.norm_std = norm_mean, norm_std
        results = {}
        for split, (vals, lens, spk) in (('train', (tr_vals, tr_lens, tr_spk)), ('val', (va_vals, va_lens, va_spk))):
            acc, n = {}, 0
            for idx in batches(len(vals), args.batch_size, False):
                preds = model(vals[idx].long(), spk[idx].long())
                for k, v in len_metrics(preds, lens[idx].to(preds.device), pad).items():
                    acc[k] = acc.get(k, 0) + float(v)
                n += int((vals[idx] != args.n_tokens).sum())
            results[split] = {k: v / max(n, 1) for k, v in acc.items()}
            results[split + '_total_mse'] = acc['MSE']
        results['train']['Loss'] = float(tot) / max(n_samples, 1)  # the train-mode loss of the steps, like the reference
        if results['val_total_mse'] < best_mse:
            torch.save(sd, out_path + '/best_model.pth')
            best_mse = results['val_total_mse']
        write_log(log, 'train', epoch, results['train'])
        write_log(log, 'val', epoch, results['val'])


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--out_path', default='checkpoints/esd', help='Path to save model and logs')
    parser.add_argument('--data_path', default='data/ESD/hubert100', help='Path to sequence data')
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
    os.makedirs(args.out_path + '/len', exist_ok=True)
    train(args.data_path, args.device, args)


if __name__ == '__main__':
    main()
