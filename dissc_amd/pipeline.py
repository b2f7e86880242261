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


def normalize_f0_(f0, lengths, mean, std, fill_median=False):
    """The vocoder data set's per-source-speaker F0 normalisation on a device batch, in place (reference
    sr/dataset.py:255-267 as sr/inference.py build_jobs restates it): voiced values (!= 0) become (f0 - mean) / std,
    evaluated in float64 like numpy does with the pickled np.float64 statistics, and with ``fill_median`` (config
    ``f0_median``) the unvoiced frames are first set to (median(voiced) - mean) / std (numpy's median: the mean of the
    two middle values).  f0 f32 [B,T]; lengths i32 [B]; mean/std f64 [B] (one pair per row)."""
    B, T = f0.shape
    valid = torch.arange(T, device=f0.device)[None, :] < lengths.to(f0.device)[:, None].long()
    voiced = (f0 != 0) & valid
    m = mean.to(f0.device, torch.float64)[:, None]
    s = std.to(f0.device, torch.float64)[:, None]
    z = ((f0.double() - m) / s).float()
    if fill_median:
        srt = torch.where(voiced, f0, torch.full_like(f0, float("inf"))).sort(dim=1).values
        nv = voiced.sum(1)
        lo = srt.gather(1, ((nv - 1).clamp(min=0) // 2)[:, None])
        hi = srt.gather(1, (nv // 2).clamp(max=T - 1)[:, None])
        med = (lo + hi) / 2          # float32, like np.median of a float32 array
        fill = ((med.double() - m) / s).float()
        has = (nv > 0)[:, None]
        f0.copy_(torch.where(voiced, z, torch.where(valid & has, fill.expand(-1, T), f0)))
    else:
        f0.copy_(torch.where(voiced, z, f0))
    return f0


class Converter:
    def __init__(self, encoder, len_model, pitch_model, generator, norm_pitch=True, n_tokens=100,
                 postprocess=True, max_batch=128, max_frames=harness.MAX_FRAMES, encode_seconds=640.0, f0_median=False):
        self.encoder, self.len_model, self.pitch_model, self.generator = encoder, len_model, pitch_model, generator
        self.norm_pitch, self.n_tokens, self.postprocess = norm_pitch, n_tokens, postprocess
        self.max_batch, self.max_frames, self.encode_seconds = max_batch, max_frames, encode_seconds
        self.f0_median = f0_median
        self.profile = None  # a dict: filled with per-stage HIP-event times (ms) of the calls that follow

    def _staging(self, b, n):
        """pinned host staging buffer for one encode batch (page-locking is expensive: allocated once, reused; the
        H2D copy below is synchronous with respect to the host, so reuse across batches is safe)"""
        need = b * n
        if getattr(self, "_pin", None) is None or self._pin.numel() < need:
            self._pin = torch.empty(need, dtype=torch.float32, pin_memory=True)
        return self._pin[:need].view(b, n)  # rows are filled (and their tails zeroed) by the caller

    def _mark(self, marks, name):
        if self.profile is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((name, ev))

    def _collect(self, marks):
        """stage times from the recorded events: the span between consecutive marks belongs to the later mark"""
        if self.profile is None or not marks:
            return
        marks[-1][1].synchronize()
        for (_, e0), (name, e1) in zip(marks[:-1], marks[1:]):
            if name.startswith("_"):
                continue
            self.profile[name] = self.profile.get(name, 0.0) + e0.elapsed_time(e1)

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _run_local(self, ids, n_samples, load, target_ids, store, f0_stats=None):
        """ids: the utterances of this rank (this round); n_samples[i] their sample counts; load(i) -> 1-D float
        array @16 kHz (host) or a device tensor, called right before utterance i's encode batch is staged (so at
        most one encode batch of audio is held).  Fills ``store`` with the waveforms of every (utterance, target),
        job id = utt_index * len(target_ids) + target slot.  f0_stats: None or (mean, std) f64 arrays indexed by
        utterance -- the vocoder config's f0_normalize step on the predicted F0."""
        dev = self.generator.device
        nt = len(target_ids)
        tgt = torch.as_tensor(list(target_ids), dtype=torch.int64)
        for m in (self.len_model, self.pitch_model):
            if m is not None and m._sd is not None:
                m._check_ids(tgt, m._sd["spk_emb.weight"].shape[0], "speaker")
        tgt_dev = tgt.to(dev)
        order = sorted(ids, key=lambda i: (-int(n_samples[i]), i))
        pos = 0
        marks = []
        while pos < len(order):  # length-sorted encode batches of <= encode_seconds of audio
            self._mark(marks, "_start")
            n0 = int(n_samples[order[pos]])
            bsz = max(1, int(self.encode_seconds * 16000 // max(n0, 1)))
            batch = order[pos:pos + bsz]
            pos += len(batch)
            b = len(batch)
            waves = [load(j) for j in batch]
            ns = np.array([len(w) for w in waves], dtype=np.int32)
            n0 = int(ns.max())  # the decoded lengths size the rows (a header may disagree with its data)
            if all(torch.is_tensor(w) and w.is_cuda for w in waves):
                # audio already resident in HBM (e.g. handed over by a device-side loader): pad on the device
                wav_dev = torch.zeros(b, n0, dtype=torch.float32, device=dev)
                for k, w in enumerate(waves):
                    wav_dev[k, :int(ns[k])] = w.reshape(-1)
            else:
                wav = self._staging(b, n0)
                for k, w in enumerate(waves):
                    wav[k, :int(ns[k])] = torch.as_tensor(np.asarray(w, dtype=np.float32).reshape(-1))
                    wav[k, int(ns[k]):] = 0.0
                wav_dev = wav.to(dev)
            del waves
            self._mark(marks, "stage_ms")
            enc = self.encoder(wav_dev, n_samples=torch.from_numpy(ns), want_dense=False)
            units = enc["units"]                       # i64 [b,T] on the device
            frames = enc["frames_dev"] if "frames_dev" in enc else enc["frames"].to(dev)  # i32 [b] (host arithmetic on n_samples, uploaded before the encoder's launches:
                                                       # an H2D copy queued BEHIND them would block the host for the whole encode)
            self._mark(marks, "encode_ms")
            # every utterance x every target, target fastest: row = k*nt + slot
            r = P.infer_batch(units.repeat_interleave(nt, 0), frames.repeat_interleave(nt),
                              tgt_dev.repeat(b), self.len_model, self.pitch_model, self.norm_pitch)
            totals = r["totals"].tolist()
            if f0_stats is not None and r["f0"] is not None:
                mean = torch.as_tensor(np.repeat(np.asarray(f0_stats[0], np.float64)[batch], nt))
                std = torch.as_tensor(np.repeat(np.asarray(f0_stats[1], np.float64)[batch], nt))
                normalize_f0_(r["f0"], r["lengths"], mean, std, self.f0_median)
            self._mark(marks, "predict_ms")
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
                if self.postprocess:
                    wav_postprocess_(y, lens * hop)
                store.add(y, np.array([totals[k] for k in gb], dtype=np.int64) * hop, [job_of_row[k] for k in gb])
            self._mark(marks, "generator_ms")
        self._collect(marks)

    def _decode(self, waves_by_job, target_ids):
        nt = len(target_ids)
        return {(j // nt, target_ids[j % nt]): w for j, w in waves_by_job.items()}

    def __call__(self, waveforms, target_ids, f0_stats=None):
        """waveforms: list of 1-D float arrays @16 kHz; target_ids: list of speaker ids (every
        utterance is converted to every target).  Returns {(utt_index, target_id): samples}."""
        store = harness.WaveStore(self.generator.device)
        marks = []
        self._run_local(list(range(len(waveforms))), [len(w) for w in waveforms], lambda i: waveforms[i],
                        list(target_ids), store, f0_stats)
        self._mark(marks, "_start")
        out = self._decode(harness.gather_store(store, store.n, store.data_floats, 0, 1), list(target_ids))
        self._mark(marks, "pack_d2h_ms")
        self._collect(marks)
        return out

    def run_sharded(self, n_samples, load, target_ids, rank=0, world_size=1, dist=None, unpack_ranks=(0,),
                    f0_stats=None, sink=None, round_floats=harness.ROUND_FLOATS, stats=None, own_rows=False,
                    overlap=None):
        """n_samples: sample count of EVERY utterance (all ranks pass the same list: it defines the
        partition); load(i) -> waveform of utterance i (called for this rank's share only, one encode batch at a
        time).  Per round: a 16-byte MAX all-reduce (the buffer geometry -- predicted durations size the outputs, so
        the ranks cannot derive it from the job list) and ONE all-gather of the waveforms; a run is one round unless a
        rank's share exceeds ``round_floats`` input samples x targets (the output length is data-dependent with a
        rhythm model; the input length is the proxy all ranks can agree on without talking) -- or, with a ``sink`` on a
        GPU, up to 4 rounds delivered by a worker thread while the next one computes (harness.Exchange).
        Returns {(utt_index, target_id): samples} on ``unpack_ranks`` (None = all), {} elsewhere -- or, with
        ``sink``, calls ``sink({(utt, target): samples})`` per round there and returns the number of waveforms
        delivered (the arrays a sink receives are views of a reused page-locked buffer: valid until it returns).
        ``own_rows=True``: every rank delivers the conversions it produced itself."""
        target_ids = list(target_ids)
        nt = max(len(target_ids), 1)
        parts = harness.lpt_shard(n_samples, world_size)
        budget = None if round_floats is None else max(1, round_floats // nt)
        ex = harness.Exchange(rank, world_size, self.generator.device, dist, unpack_ranks, own_rows, sink, stats,
                              decode=lambda got: self._decode(got, target_ids), overlap=overlap)
        if ex.plan_cut:  # (agreed between the ranks) input frames (320 samples) as the length unit of the overlap rule
            b = harness.overlap_budget([n // 320 for n in n_samples], parts, None,
                                       cap=max(1, harness.OVERLAP_CAP_FLOATS // (320 * nt)))
            if b is not None:  # (a list: the tapered cut)
                budget = [v * 320 if budget is None else min(budget, v * 320) for v in b]
        try:
            for shares in harness.plan_rounds(n_samples, parts, budget):
                store = harness.WaveStore(self.generator.device)
                self._run_local(shares[rank], n_samples, load, target_ids, store, f0_stats)
                n_cap = max(len(p) for p in shares) * len(target_ids)
                _, data_cap = harness.agree_geometry(store.n, store.data_floats, world_size, self.generator.device, dist)
                ex.submit(store, n_cap, data_cap)
                store.clear()
            return ex.finish()
        finally:
            ex.close()  # (idempotent: a run that fails half-way must not leave the delivery thread behind)
