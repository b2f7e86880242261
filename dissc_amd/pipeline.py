"""In-memory DISSC conversion: waveforms -> units -> (rhythm, pitch) -> waveforms on one GPU,
without the JSON-lines round trips between the reference's three scripts.

    conv = Converter(encoder, len_model, pitch_model, generator)
    wavs = conv(waveforms, target_ids)          # list of 1-D float32 arrays, one per (utt, target)

The file-based entry points (data/encode.py -> infer.py -> sr/inference.py) remain the
format-compatible path; this is the same kernels chained on device-resident tensors.
"""
import numpy as np
import torch

from . import predictors as P
from .generator import wav_postprocess_


class Converter:
    def __init__(self, encoder, len_model, pitch_model, generator, norm_pitch=True, n_tokens=100,
                 postprocess=True, max_batch=32, max_frames=32 * 500):
        self.encoder, self.len_model, self.pitch_model, self.generator = encoder, len_model, pitch_model, generator
        self.norm_pitch, self.n_tokens, self.postprocess = norm_pitch, n_tokens, postprocess
        self.max_batch, self.max_frames = max_batch, max_frames

    @torch.no_grad()
    def __call__(self, waveforms, target_ids):
        """waveforms: list of 1-D float arrays @16 kHz; target_ids: list of speaker ids (every
        utterance is converted to every target).  Returns {(utt_index, target_id): samples}."""
        dev = self.generator.device
        order = sorted(range(len(waveforms)), key=lambda i: -len(waveforms[i]))
        units = {}
        i = 0
        while i < len(order):  # length-sorted encode batches of <= ~640 s of audio
            n0 = len(waveforms[order[i]])
            bsz = max(1, int(640 * 16000 // max(n0, 1)))
            batch = order[i:i + bsz]
            i += len(batch)
            wav = np.zeros((len(batch), n0), dtype=np.float32)
            ns = np.zeros(len(batch), dtype=np.int32)
            for k, j in enumerate(batch):
                wav[k, :len(waveforms[j])] = waveforms[j]
                ns[k] = len(waveforms[j])
            out = self.encoder(torch.from_numpy(wav), n_samples=torch.from_numpy(ns), want_dense=False)
            for k, j in enumerate(batch):
                units[j] = out["units"][k, :int(out["frames"][k])]
        jobs = [(j, t) for j in range(len(waveforms)) for t in target_ids]
        res = P.infer_samples([units[j].cpu() for j, _ in jobs], [t for _, t in jobs], self.len_model,
                              self.pitch_model, norm_pitch=self.norm_pitch, n_tokens=self.n_tokens, device=dev)
        from .harness import make_batches
        lengths = [len(r[0]) for r in res]
        out = {}
        for batch in make_batches(list(range(len(jobs))), lengths, self.max_batch, self.max_frames):
            B, T = len(batch), max(lengths[k] for k in batch)
            if T == 0:
                for k in batch:
                    out[jobs[k]] = np.zeros(0, np.float32)
                continue
            code = np.zeros((B, T), np.int64)
            f0 = np.zeros((B, 1, T), np.float32)
            lens = np.zeros(B, np.int32)
            spk = np.zeros((B, 1), np.int64)
            for r, k in enumerate(batch):
                u, f, _ = res[k]
                code[r, :len(u)] = u
                f0[r, 0, :len(u)] = f if f is not None else 0.0
                lens[r] = len(u)
                spk[r, 0] = jobs[k][1]
            y = self.generator(code=torch.from_numpy(code), f0=torch.from_numpy(f0), spkr=torch.from_numpy(spk),
                               lengths=torch.from_numpy(lens))
            hop = y.shape[-1] // T
            if self.postprocess:
                wav_postprocess_(y, torch.from_numpy(lens * hop))
            yc = y.cpu().numpy()
            for r, k in enumerate(batch):
                out[jobs[k]] = yc[r, 0, :lengths[k] * hop].copy()
        return out
