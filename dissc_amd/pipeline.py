"""In-memory DISSC conversion: waveforms -> units -> (rhythm, pitch) -> waveforms, device-resident
and data-parallel over ranks (SURVEY.md 8f N1 + 8e; BASELINE.json configs[1], [3], [4]).

    conv = Converter(encoder, len_model, pitch_model, generator)
    wavs = conv(waveforms, target_ids)                       # one GPU: {(utt, target): samples}
    wavs = conv.run_sharded(n_samples, load, target_ids, rank, world, dist)   # one rank per GPU

The reference chains three scripts through JSON-lines files (data/encode.py:40-41 ->
infer.py:42-44 -> sr/dataset.py:107-122) and converts one (utterance, target) at a time.  Here the
unit sequences, predicted durations, F0 and the generator's inputs never leave HBM between the
stages: the k-means units of an encode batch are replicated per target on the device, run through
``predictors.infer_batch`` (ONE host sync per encode batch: the predicted lengths, which size the
generator batches), gathered into length-sorted generator batches with device index ops, and the
post-processed waveforms are packed on the device (``harness.WaveStore``).  The only host traffic
is the input audio, a few hundred bytes of batch indices, and the final packed waveform buffer.

Multi-GPU: utterances are LPT-sharded over ranks by sample count (every rank derives the same
partition from the list of lengths); each rank encodes, predicts and resynthesises its share for
all targets; ONE all-gather returns the waveforms (preceded by a 16-byte MAX all-reduce to agree on
the row length when a rhythm model decides the output lengths).  The file-based entry points
(data/encode.py -> infer.py -> sr/inference.py) remain the format-compatible path.
"""
import numpy as np
import torch

from . import harness
from . import predictors as P
from .generator import wav_postprocess_


class Converter:
    def __init__(self, encoder, len_model, pitch_model, generator, norm_pitch=True, n_tokens=100,
                 postprocess=True, max_batch=32, max_frames=32 * 500, encode_seconds=640.0):
        self.encoder, self.len_model, self.pitch_model, self.generator = encoder, len_model, pitch_model, generator
        self.norm_pitch, self.n_tokens, self.postprocess = norm_pitch, n_tokens, postprocess
        self.max_batch, self.max_frames, self.encode_seconds = max_batch, max_frames, encode_seconds

    def _staging(self, b, n):
        """pinned host staging buffer for one encode batch (page-locking is expensive: allocated once, reused; the
        H2D copy below is synchronous with respect to the host, so reuse across batches is safe)"""
        need = b * n
        if getattr(self, "_pin", None) is None or self._pin.numel() < need:
            self._pin = torch.empty(need, dtype=torch.float32, pin_memory=True)
        return self._pin[:need].view(b, n)  # rows are filled (and their tails zeroed) by the caller

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _run_local(self, waves, target_ids, store):
        """waves: {utt_index: 1-D float array @16 kHz} of this rank.  Fills ``store`` with the
        waveforms of every (utterance, target), job id = utt_index * len(target_ids) + target slot.
        Returns the longest waveform (samples)."""
        dev = self.generator.device
        nt = len(target_ids)
        hop = None
        l_max = 0
        tgt = torch.as_tensor(list(target_ids), dtype=torch.int64)
        for m in (self.len_model, self.pitch_model):
            if m is not None and m._sd is not None:
                m._check_ids(tgt, m._sd["spk_emb.weight"].shape[0], "speaker")
        tgt_dev = tgt.to(dev)
        order = sorted(waves, key=lambda i: (-len(waves[i]), i))
        pos = 0
        while pos < len(order):  # length-sorted encode batches of <= encode_seconds of audio
            n0 = len(waves[order[pos]])
            bsz = max(1, int(self.encode_seconds * 16000 // max(n0, 1)))
            batch = order[pos:pos + bsz]
            pos += len(batch)
            b = len(batch)
            ns = np.array([len(waves[j]) for j in batch], dtype=np.int32)
            if all(torch.is_tensor(waves[j]) and waves[j].is_cuda for j in batch):
                # audio already resident in HBM (e.g. handed over by a device-side loader): pad on the device
                wav_dev = torch.zeros(b, n0, dtype=torch.float32, device=dev)
                for k, j in enumerate(batch):
                    wav_dev[k, :int(ns[k])] = waves[j].reshape(-1)
            else:
                wav = self._staging(b, n0)
                for k, j in enumerate(batch):
                    wav[k, :int(ns[k])] = torch.as_tensor(np.asarray(waves[j], dtype=np.float32).reshape(-1))
                    wav[k, int(ns[k]):] = 0.0
                wav_dev = wav.to(dev)
            enc = self.encoder(wav_dev, n_samples=torch.from_numpy(ns), want_dense=False)
            units = enc["units"]                       # i64 [b,T] on the device
            frames = enc["frames"].to(dev)             # i32 [b] (computed from n_samples on the host)
            # every utterance x every target, target fastest: row = k*nt + slot
            r = P.infer_batch(units.repeat_interleave(nt, 0), frames.repeat_interleave(nt),
                              tgt_dev.repeat(b), self.len_model, self.pitch_model, self.norm_pitch)
            totals = r["totals"].tolist()
            spk_rows = tgt_dev.repeat(b)
            job_of_row = [batch[row // nt] * nt + row % nt for row in range(b * nt)]
            for gb in harness.make_batches(list(range(b * nt)), totals, self.max_batch, self.max_frames):
                T = max(totals[k] for k in gb)
                if T == 0:
                    store.add_empty([job_of_row[k] for k in gb])
                    continue
                idx = torch.as_tensor(gb, dtype=torch.int64).to(dev)
                code = r["units"].index_select(0, idx)[:, :T].contiguous()
                if r["f0"] is not None:
                    f0 = r["f0"].index_select(0, idx)[:, :T].contiguous().unsqueeze(1)
                else:
                    f0 = torch.zeros(len(gb), 1, T, dtype=torch.float32, device=dev)
                lens = r["lengths"].index_select(0, idx)
                y = self.generator(code=code, f0=f0, spkr=spk_rows.index_select(0, idx).view(-1, 1), lengths=lens)
                hop = y.shape[-1] // T
                nsamp = lens * hop
                if self.postprocess:
                    wav_postprocess_(y, nsamp)
                store.add(y, nsamp, [job_of_row[k] for k in gb])
                l_max = max(l_max, T * hop)
        return l_max

    def _decode(self, waves_by_job, target_ids):
        nt = len(target_ids)
        return {(j // nt, target_ids[j % nt]): w for j, w in waves_by_job.items()}

    def __call__(self, waveforms, target_ids):
        """waveforms: list of 1-D float arrays @16 kHz; target_ids: list of speaker ids (every
        utterance is converted to every target).  Returns {(utt_index, target_id): samples}."""
        store = harness.WaveStore(self.generator.device)
        l_max = self._run_local(dict(enumerate(waveforms)), list(target_ids), store)
        return self._decode(harness.gather_store(store, store.n, l_max, 0, 1), list(target_ids))

    def run_sharded(self, n_samples, load, target_ids, rank=0, world_size=1, dist=None, unpack_ranks=(0,)):
        """n_samples: sample count of EVERY utterance (all ranks pass the same list: it defines the
        partition); load(i) -> waveform of utterance i (called for this rank's share only).
        Returns {(utt_index, target_id): samples} on ``unpack_ranks`` (None = all), {} elsewhere."""
        target_ids = list(target_ids)
        parts = harness.lpt_shard(n_samples, world_size)
        store = harness.WaveStore(self.generator.device)
        l_loc = self._run_local({i: load(i) for i in parts[rank]}, target_ids, store)
        n_max = max(len(p) for p in parts) * len(target_ids)
        _, l_max = harness.agree_geometry(store.n, l_loc, world_size, self.generator.device, dist)
        out = harness.gather_store(store, n_max, l_max, rank, world_size, dist, unpack_ranks)
        return self._decode(out, target_ids)
