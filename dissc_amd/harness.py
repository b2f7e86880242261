"""Data-parallel harness: shard jobs over ranks, batch them, and return every waveform to rank 0
with ONE all-gather per round.

Replaces the reference's ``multiprocessing.Pool(8)`` of B=1 workers writing their own files
(reference sr/inference.py:288-292,351-354): one process per GPU (torchrun env), each rank
batches its share through ``dissc_amd.CodeGenerator`` and the decoded waveforms of all ranks are
exchanged with a single ``all_gather_into_tensor`` of a packed RAGGED fp32 buffer (header + row table
+ rows back to back, each padded to 16 bytes only; layout and the pack kernel: include/dissc_hip.h
``dissc_pack_rows``).  Utterances are independent, so there is no other collective on the data path.

A run is ONE round -- one all-gather -- unless a rank's share exceeds ``round_floats`` (default 2^28
floats = 1 GiB = 4.6 hours of 16 kHz audio per rank): then every rank cuts its share into the same number
of rounds, and each round is packed, gathered, handed to ``sink`` and FREED before the next one starts, so the
resident footprint is bounded by the round and a failure loses one round at most.  With a ``sink`` on a GPU a
run is cut into up to 4 rounds anyway and ``Exchange`` delivers round k (wait for the gather, device-to-host
copy on a copy stream, unpack, sink) on a worker thread while round k + 1 computes: the serial tail of a run at
N ranks is the delivery of its last round only.  Who delivers: rank 0 everything, or (``own_rows``) every rank
the rows it decoded itself (the round's all-gather then carries the row tables only: ``Exchange``).

Everything here is host logic; it runs on CPU tensors with the gloo backend in the tests (there the
rows are packed with torch copies) and on CUDA tensors over RCCL/xGMI in production.  A non-None ``dist``
means "a process group exists": the collective then runs at ANY world size (world_size 1 included --
``DISSC_FORCE_DIST=1`` in the CLIs -- which is how RCCL is exercised on a one-GPU box).
"""
import os

import numpy as np
import torch

HDR = 4               # header floats (16 B)
ENT = 4               # table floats per row (16 B): job id, sample count, offset lo, offset hi
ROUND_FLOATS = 1 << 28
MAX_FRAMES = 48 * 500  # padded frames (B x Tmax) per generator call, see make_batches


def lpt_shard(lengths, world_size):
    """Longest-processing-time greedy assignment of jobs to ranks.  Deterministic (ties broken
    by job index), so every rank computes the same partition from the manifest alone.
    Returns list[world_size] of job-index lists."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world_size
    parts = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += int(lengths[i])
    return parts


def imbalance(lengths, parts):
    """max / mean of the per-rank loads of a partition (1.0 = perfectly even)."""
    loads = [sum(int(lengths[i]) for i in p) for p in parts]
    mean = sum(loads) / max(len(loads), 1)
    return max(loads) / mean if mean > 0 else 1.0


def make_batches(job_ids, lengths, max_batch=128, max_frames=MAX_FRAMES):
    """Length-sorted batches of at most ``max_batch`` jobs and ``max_frames`` padded frames
    (Tmax * B), so padding waste stays small and the workspace bounded (it grows with the PADDED frames, about 0.4 MB
    each; the kernels themselves skip what lies beyond an utterance's end).  24 000 = the 32 x 10 s of the headline batch
    with room for a rhythm model that stretches some of its utterances by half -- at 16 000 such a batch was cut into
    25 + 7 utterances and the small second call cost 8 % of the conversion; short utterances ride up to 128 to a batch,
    which keeps the launch grids full (1 024 jobs of 2-5 s: 462 ms at 64 x 16 000, 447 ms at 128 x 24 000,
    `tools/strong_ab.py`)."""
    ids = sorted(job_ids, key=lambda i: (-int(lengths[i]), i))
    batches, cur = [], []
    for i in ids:
        tmax = int(lengths[cur[0]]) if cur else int(lengths[i])
        if cur and (len(cur) >= max_batch or tmax * (len(cur) + 1) > max_frames):
            batches.append(cur)
            cur = []
        cur.append(i)
    if cur:
        batches.append(cur)
    return batches


def plan_rounds(lengths, parts, budget=None):
    """Cut every rank's share (longest first) into rounds of at most ``budget`` length units (a round
    always takes at least one job).  Every rank derives the same plan; the number of rounds is the
    maximum over ranks (a rank that runs out contributes empty rounds, it still joins the collectives).
    Returns list[round] of list[world_size] of job-index lists.  ``budget`` None = one round; a sequence = the
    budgets of rounds 0, 1, ... (``overlap_budget``'s tapered cut; its last entry serves any further round)."""
    seq = None if budget is None else ([int(b) for b in budget] if isinstance(budget, (list, tuple)) else [int(budget)])
    per_rank = []
    for p in parts:
        ids = sorted(p, key=lambda i: (-int(lengths[i]), i))
        if seq is None:
            per_rank.append([ids])
            continue
        rounds, cur, tot = [], [], 0
        for i in ids:
            n = int(lengths[i])
            if cur and tot + n > seq[min(len(rounds), len(seq) - 1)]:
                rounds.append(cur)
                cur, tot = [], 0
            cur.append(i)
            tot += n
        rounds.append(cur)
        per_rank.append(rounds)
    n_rounds = max((len(r) for r in per_rank), default=1)
    return [[r[k] if k < len(r) else [] for r in per_rank] for k in range(n_rounds)]


def _r4(n):
    return (int(n) + 3) // 4 * 4


def pack_geometry(lengths, parts, hop):
    """(n_cap, data_cap) of the exchange buffer -- the most rows and the most (16-byte padded) data floats
    any rank packs; derived from the global job list, so all ranks agree without a collective."""
    n_cap = max((len(p) for p in parts), default=0)
    data_cap = max((sum(_r4(int(lengths[i]) * hop) for i in p) for p in parts), default=0)
    return n_cap, data_cap


def buffer_floats(n_cap, data_cap):
    return HDR + ENT * int(n_cap) + int(data_cap)


class WaveStore:
    """The decoded waveforms of one rank, kept on the device as the generator produced them:
    a list of batches (wav [B, ld], sample counts, job ids).  Sample counts live on the HOST (both
    callers know them without a sync): the row offsets of the ragged pack are their prefix sum."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.batches = []
        self.n = 0
        self.data_floats = 0  # sum of the 16-byte padded sample counts

    def add(self, wav, n_samples, job_ids):
        """wav f32 [B,1,L] or [B,L]; n_samples int [B] (host preferred; a device tensor costs a sync);
        job_ids: ints [B]"""
        w2 = wav.view(wav.shape[0], -1)
        if torch.is_tensor(n_samples):
            n_samples = n_samples.cpu().numpy()
        ns = np.minimum(np.asarray(n_samples, dtype=np.int64).reshape(-1), w2.shape[1]).astype(np.int32)
        ids = np.asarray(job_ids, dtype=np.int32).reshape(-1)
        assert ns.size == w2.shape[0] == ids.size
        self.batches.append((w2, ns, ids))
        self.n += w2.shape[0]
        self.data_floats += int(((ns.astype(np.int64) + 3) // 4 * 4).sum())

    def add_empty(self, job_ids):
        k = len(job_ids)
        if k:
            self.add(torch.zeros(k, 4, dtype=torch.float32, device=self.device), np.zeros(k, np.int32), job_ids)

    def clear(self):
        self.batches, self.n, self.data_floats = [], 0, 0

    def pack(self, n_cap, data_cap):
        """-> f32 [4 + 4*n_cap + data_cap] on the device (layout: include/dissc_hip.h)."""
        if self.n > n_cap or self.data_floats > data_cap:
            raise ValueError(f"{self.n} waveforms / {self.data_floats} floats do not fit "
                             f"{n_cap} rows / {data_cap} floats")
        ns = np.concatenate([b[1] for b in self.batches]) if self.batches else np.zeros(0, np.int32)
        ids = np.concatenate([b[2] for b in self.batches]) if self.batches else np.zeros(0, np.int32)
        slots = (ns.astype(np.int64) + 3) // 4 * 4
        offs = np.cumsum(slots) - slots
        head = np.zeros((1 + n_cap, 4), dtype=np.int32)
        head[0, 0] = self.n
        head[0, 2:4] = np.array([self.data_floats], dtype=np.int64).view(np.int32)
        head[1:, 0] = -1
        head[1:1 + self.n, 0] = ids
        head[1:1 + self.n, 1] = ns
        head[1:1 + self.n, 2:4] = offs.astype(np.int64).view(np.int32).reshape(-1, 2)
        tbl = HDR + ENT * n_cap
        buf = torch.empty(buffer_floats(n_cap, data_cap), dtype=torch.float32, device=self.device)
        if self.device.type == "cuda":
            from ._lib import check, current_stream_ptr, lib
            # header, sample counts and offsets go over from ONE page-locked block with non-blocking copies: a
            # pageable copy_ is ordered behind everything on the stream and blocks the host until the round's
            # kernels are done -- the caller could not start batching the next round meanwhile
            n_h, n_r = head.size, max(self.n, 1)
            stage = torch.empty(2 * n_r + n_h + n_r, dtype=torch.int32, pin_memory=True)  # [offsets i64 | header | counts]
            sn = stage.numpy()
            sn[:2 * self.n] = offs.astype(np.int64).view(np.int32)
            sn[2 * n_r:2 * n_r + n_h] = head.reshape(-1)
            sn[2 * n_r + n_h:2 * n_r + n_h + self.n] = ns
            d_stage = stage.to(self.device, non_blocking=True)
            d_off = d_stage[:2 * n_r].view(torch.int64)   # at offset 0: 8-byte aligned
            buf[:tbl].copy_(d_stage[2 * n_r:2 * n_r + n_h].view(torch.float32))
            d_ns = d_stage[2 * n_r + n_h:]
            data = buf[tbl:]
            assert data.data_ptr() % 16 == 0
            with torch.cuda.device(self.device):
                st = current_stream_ptr(self.device)
                row = 0
                for w2, bns, _ in self.batches:
                    B = w2.shape[0]
                    check(lib.dissc_pack_rows(w2.data_ptr(), w2.stride(0), d_ns[row:].data_ptr(),
                                              d_off[row:].data_ptr(), B, int(bns.max()) if B else 0,
                                              data.data_ptr(), st), "dissc_pack_rows")
                    row += B
        else:  # gloo/CPU rehearsal of the same layout (tests)
            buf[:tbl].copy_(torch.from_numpy(head.reshape(-1)).view(torch.float32))
            buf[tbl:].zero_()
            row = 0
            for w2, bns, _ in self.batches:
                for k in range(w2.shape[0]):
                    o, n = tbl + int(offs[row]), int(bns[k])
                    buf[o:o + n] = w2[k, :n]
                    row += 1
        return buf


def pack_waves(waves, job_ids, n_cap, data_cap, device):
    """waves: list of 1-D float tensors; -> packed exchange buffer of one rank."""
    st = WaveStore(device)
    for w, j in zip(waves, job_ids):
        w = w.reshape(1, -1).to(st.device, torch.float32)
        n = w.shape[1]
        if n == 0:
            st.add_empty([j])
        else:
            st.add(w.contiguous(), [n], [j])
    return st.pack(n_cap, data_cap)


_PIN = {"buf": None}
_PIN_MAX_FLOATS = (64 << 20) // 4  # page-locking costs ~0.2 ms per MB once; larger buffers go through in chunks


def _pinned(n_floats):
    """page-locked staging (cached: page-locking costs more than the copy itself), at most 64 MB"""
    need = max(1, min(int(n_floats), _PIN_MAX_FLOATS))
    if _PIN["buf"] is None or _PIN["buf"].numel() < need:
        _PIN["buf"] = None
        _PIN["buf"] = torch.empty(need, dtype=torch.float32, pin_memory=True)
    return _PIN["buf"][:need]


def _to_host(t):
    """1-D device f32 tensor -> numpy copy, through the cached page-locked staging buffer (a pageable ``.cpu()`` of the
    20 MB of a 32-utterance batch takes 1.7 ms, this 0.7 ms).  Above 32 MB the buffer works as two halves: the D2H of
    chunk i + 1 runs while the host copies chunk i out of the other half (226 MB: 34 -> 27 ms)."""
    n = t.numel()
    out = np.empty(n, dtype=np.float32)
    stage = _pinned(n)
    if n <= stage.numel() and n <= _PIN_MAX_FLOATS // 2:
        stage[:n].copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        out[:] = stage[:n].numpy()
        return out
    cap = stage.numel() // 2
    halves = (stage[:cap], stage[cap:2 * cap])
    chunks = [(o, min(cap, n - o)) for o in range(0, n, cap)]
    # copies and events both go to t's device's current stream: on a non-current device a bare Event.record() would
    # land on another device's stream and order nothing
    with torch.cuda.device(t.device):
        st = torch.cuda.current_stream(t.device)
        events = (torch.cuda.Event(), torch.cuda.Event())
        o, k = chunks[0]
        halves[0][:k].copy_(t[o:o + k], non_blocking=True)
        events[0].record(st)
        for i, (o, k) in enumerate(chunks):
            if i + 1 < len(chunks):
                o2, k2 = chunks[i + 1]
                halves[(i + 1) & 1][:k2].copy_(t[o2:o2 + k2], non_blocking=True)
                events[(i + 1) & 1].record(st)
            events[i & 1].synchronize()
            out[o:o + k] = halves[i & 1][:k].numpy()
    return out


_PIN_ROUND = {}
_PIN_ROUND_MAX_FLOATS = (2 << 30) // 4  # 2 GiB


def _pinned_round(n_floats, slot=0):
    """A page-locked buffer for a whole round's waveforms (grown on demand, kept: page-locking 226 MB costs 45 ms), or
    None when the round is larger than 2 GiB.  Two slots: while a sink still reads round k out of one, round k + 1
    lands in the other (Exchange)."""
    n = int(n_floats)
    if n > _PIN_ROUND_MAX_FLOATS:
        return None
    if _PIN_ROUND.get(slot) is None or _PIN_ROUND[slot].numel() < n:
        _PIN_ROUND[slot] = None
        _PIN_ROUND[slot] = torch.empty(max(n, 1), dtype=torch.float32, pin_memory=True)
    return _PIN_ROUND[slot]


def unpack_waves(gathered, n_cap, copy=False, stats=None, transient=False, ranks=None, slot=0):
    """gathered: f32 [world * (4 + 4*n_cap + data_cap)] or [world, ...] (any device) -> {job_id: 1-D float32
    numpy array}.  From a device buffer only what the tables say is in use comes over: the world headers +
    tables first (a few KB), then each rank's used data region through the page-locked staging buffer.
    A host buffer is viewed in place unless ``copy``.  ``transient``: the caller consumes the arrays before the next
    call (a ``sink`` that writes the files of a round) -- they are then views of ONE cached page-locked buffer the
    device copies straight into, which saves the host memcpy out of the staging buffer (22 ms per 226 MB); ``slot``
    picks which of the two cached buffers.  ``ranks``: only the regions packed by these source ranks are read
    (None = all) -- the others are neither copied nor unpacked.  Device work goes to the CURRENT stream of the buffer's
    device (the Exchange worker calls this under its own copy stream)."""
    tbl = HDR + ENT * n_cap
    g = gathered.reshape(-1)
    per = None
    out = {}
    # the buffer length per rank is not stored: recover it from the caller's shape when 2-D, else it is one rank
    if gathered.dim() == 2:
        world, per = gathered.shape
    else:
        world, per = 1, g.numel()
    if per < tbl:
        raise ValueError("exchange buffer shorter than its table")
    g2 = g.view(world, per)
    heads = g2[:, :tbl]
    heads = (heads.cpu() if heads.is_cuda else heads).contiguous().numpy().view(np.int32).reshape(world, 1 + n_cap, 4)
    moved = 0
    useds = [int(heads[r, 0, 2:4].copy().view(np.int64)[0]) for r in range(world)]
    rows = [int(heads[r, 0, 0]) for r in range(world)]
    for r in range(world):
        if rows[r] < 0 or rows[r] > n_cap or useds[r] < 0 or useds[r] > per - tbl:
            raise ValueError(f"rank {r}: corrupt exchange header (rows {rows[r]}, data floats {useds[r]})")
    if ranks is not None:
        keep = set(int(r) for r in ranks)
        useds = [u if r in keep else 0 for r, u in enumerate(useds)]
        rows = [n if r in keep else 0 for r, n in enumerate(rows)]
    big = None
    if transient and g2.is_cuda and all(0 <= u <= per - tbl for u in useds):
        big = _pinned_round(sum(useds), slot)
        if big is not None:  # every rank's region straight into its slice of the round buffer, one synchronisation
            o = 0
            for r in range(world):
                if useds[r]:
                    big[o:o + useds[r]].copy_(g2[r, tbl:tbl + useds[r]], non_blocking=True)
                o += useds[r]
            torch.cuda.current_stream(g2.device).synchronize()
    big_off = 0
    for r in range(world):
        n_rows = rows[r]
        used = useds[r]
        if big is not None:
            data_big = big[big_off:big_off + used].numpy()
            big_off += used
        if n_rows == 0:
            continue
        if big is not None:
            data = data_big
            moved += used
            own = True
        elif g2.is_cuda:
            data = _to_host(g2[r, tbl:tbl + used])
            moved += used
            own = True
        else:
            data = g2[r, tbl:tbl + used].numpy()
            own = False
        ent = heads[r, 1:1 + n_rows]
        offs = ent[:, 2:4].copy().view(np.int64).reshape(-1)
        for k in range(n_rows):
            w = data[int(offs[k]):int(offs[k]) + int(ent[k, 1])]
            out[int(ent[k, 0])] = w.copy() if (copy and not own) else w
    if stats is not None:
        stats["d2h_floats"] = stats.get("d2h_floats", 0) + moved + (world * tbl if g2.is_cuda else 0)
    return out


def gather_store(store, n_cap, data_cap, rank, world_size, dist=None, unpack_ranks=(0,), stats=None, transient=False,
                 src_ranks=None, tables_only=False):
    """The single collective of a round, synchronously.  Returns {job_id: samples} on the ranks in ``unpack_ranks``
    (None = every rank), {} elsewhere -- only the consumers pay the device-to-host copy (``src_ranks``: and only for
    the regions those ranks packed).  The collective runs whenever a process group is given (``dist`` not None),
    also at world_size 1.  ``tables_only`` (own-rows delivery, the caller unpacks its OWN buffer): the collective carries
    the header + row table of every rank (16 B per row) instead of the samples."""
    buf = store.pack(n_cap, data_cap)
    lean = tables_only and dist is not None
    _exchange_stats(stats, store, buf, dist, HDR + ENT * n_cap if lean else None)
    if lean:
        # own-rows delivery: no rank reads another rank's samples, so only the row TABLES travel (who decoded what, how long)
        tables = torch.empty(world_size * (HDR + ENT * n_cap), dtype=torch.float32, device=buf.device)
        dist.all_gather_into_tensor(tables, buf[:HDR + ENT * n_cap].contiguous())
        _check_tables(tables.view(world_size, -1), n_cap, stats)
        buf, src_ranks = buf.view(1, -1), None
    elif dist is not None:
        out = torch.empty(world_size * buf.numel(), dtype=torch.float32, device=buf.device)
        dist.all_gather_into_tensor(out, buf)
        buf = out.view(world_size, -1)
    else:
        buf = buf.view(1, -1)
    if unpack_ranks is not None and rank not in unpack_ranks:
        return {}
    return unpack_waves(buf, n_cap, stats=stats, transient=transient, ranks=src_ranks)


class Exchange:
    """The rounds of one run as a pipeline: pack -> all-gather -> device-to-host -> unpack -> ``sink``.

    ``submit(store, n_cap, data_cap)`` packs a round on the caller's stream and starts its all-gather ASYNCHRONOUSLY;
    with a ``sink`` on a GPU the rest of the round -- waiting for the gather, the copy of the wanted regions into one of
    two page-locked round buffers on a copy stream of its own, the unpack and the sink (rank 0 writing files) -- runs on
    a worker thread, so the caller goes straight on to batch and launch the next round: at N ranks the serial tail of a
    run is the delivery of its LAST round only, not of all of it (reference sr/inference.py:288-292,351-357 -- there
    every pool worker writes its own files; here one collective per round brings them together).  At most two rounds
    are in flight (two page-locked slots); a third ``submit`` waits for the oldest to be delivered.  Without a sink
    (results returned as a dict) or on the CPU the rounds are delivered synchronously, as before.

    Who delivers what: ``unpack_ranks`` (default rank 0; None = every rank) receive EVERY waveform of the round;
    ``own_rows=True`` instead makes every rank deliver exactly the rows it packed itself, which spreads the
    device-to-host copies and the file writes over the ranks.  Nobody then reads another rank's samples, so the round's
    collective carries the row TABLES only (header + 16 B per waveform: every rank still learns what every rank delivered,
    ``stats['rows_all_ranks']``) and each rank unpacks its own buffer; ``DISSC_OWN_ROWS_GATHER=full`` sends the whole
    buffers as rounds 4 and earlier did (N x the payload over xGMI and into HBM, read by nobody).
    ``decode`` maps the {job id: samples} of a round to what the sink / the result dict should see.
    Collectives are only ever ISSUED from the caller's thread, in the same order on every rank; the worker only waits."""

    def __init__(self, rank, world_size, device, dist=None, unpack_ranks=(0,), own_rows=False, sink=None, stats=None,
                 decode=None, overlap=None):
        self.rank, self.world, self.device, self.dist = int(rank), int(world_size), torch.device(device), dist
        self.own_rows = bool(own_rows)
        self.delivers = self.own_rows or unpack_ranks is None or self.rank in unpack_ranks
        self.src_ranks = [self.rank] if self.own_rows else None
        # own_rows: nobody reads another rank's samples, so the round's collective carries the row tables only (16 B per
        # waveform) unless DISSC_OWN_ROWS_GATHER=full asks for the whole buffers (N x the payload over xGMI, for nothing)
        self.lean = self.own_rows and dist is not None and os.environ.get("DISSC_OWN_ROWS_GATHER", "tables") != "full"
        self.sink, self.decode, self.stats = sink, decode, stats
        cuda = self.device.type == "cuda"
        if overlap is None:
            overlap = cuda and sink is not None and os.environ.get("DISSC_EXCHANGE_OVERLAP", "1") != "0"
        self.overlap = bool(overlap) and sink is not None
        # Whether the run is CUT into tapered rounds is part of the round plan, and the plan must be the same on every rank (one
        # all-gather per round): a rank that passes no sink (rank != 0 with unpack_ranks=(0,)) would plan one round while rank 0
        # plans four, and the run would hang.  So the ranks agree once: cut if ANY rank delivers overlapped (collectives are
        # issued from the caller's thread only, here as everywhere).
        self.plan_cut = self.overlap
        if dist is not None and self.world > 1:
            flag = torch.tensor([int(self.overlap)], dtype=torch.int32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            self.plan_cut = bool(int(flag.item()))
        self.result, self.delivered, self.k = {}, 0, 0
        self.error = None
        self._pending = []   # (work, buf, out) of rounds this rank does not deliver: kept until the collective is done
        self._thread = self._q = self._slots = self._side = None
        self.t = {"pack_s": 0.0, "submit_wait_s": 0.0, "gather_wait_s": 0.0, "unpack_s": 0.0, "sink_s": 0.0}
        if self.overlap and self.delivers:
            import queue
            import threading
            self._q = queue.Queue()
            self._slots = threading.Semaphore(2)
            if cuda:
                with torch.cuda.device(self.device):
                    self._side = torch.cuda.Stream(self.device)
            self._thread = threading.Thread(target=self._run, name="dissc-exchange", daemon=True)
            self._thread.start()

    # -- worker ------------------------------------------------------------------------------------------------
    def _deliver(self, k, work, ev, buf, n_cap, tables=None):
        import time
        t0 = time.perf_counter()
        if self.device.type == "cuda":
            with torch.cuda.device(self.device), torch.cuda.stream(self._side):
                if work is not None:
                    work.wait()               # RCCL: the copy stream waits for the collective; gloo: the host does
                if ev is not None:
                    self._side.wait_event(ev)
                self._side.synchronize()      # split "waiting for the round" from "moving it"
                t1 = time.perf_counter()
                if self.lean:
                    _check_tables(tables.view(self.world, -1), n_cap, self.stats)
                got = unpack_waves(buf, n_cap, stats=self.stats, transient=True,
                                   ranks=None if self.lean else self.src_ranks, slot=k & 1)
        else:
            if work is not None:
                work.wait()
            t1 = time.perf_counter()
            if self.lean:
                _check_tables(tables.view(self.world, -1), n_cap, self.stats)
            got = unpack_waves(buf, n_cap, stats=self.stats, transient=True, ranks=None if self.lean else self.src_ranks)
        t2 = time.perf_counter()
        if self.decode is not None:
            got = self.decode(got)
        if got:
            self.sink(got)
        self.delivered += len(got)
        t3 = time.perf_counter()
        self.t["gather_wait_s"] += t1 - t0
        self.t["unpack_s"] += t2 - t1
        self.t["sink_s"] += t3 - t2

    def _run(self):
        while True:
            item = self._q.get()
            if item is None:
                return
            try:
                if self.error is None:
                    self._deliver(*item[:5], tables=item[6] if self.lean else None)
            except BaseException as e:  # noqa: BLE001  (handed to the caller's thread by submit / finish)
                self.error = e
            finally:
                del item
                self._slots.release()

    # -- caller ------------------------------------------------------------------------------------------------
    def _raise(self):
        if self.error is not None:
            e, self.error = self.error, None
            raise e

    def submit(self, store, n_cap, data_cap):
        import time
        if not (self.overlap and (self.dist is not None or self.delivers)):
            # synchronous round (no sink, CPU, or overlap switched off): pack, gather, unpack, deliver
            got = gather_store(store, n_cap, data_cap, self.rank, self.world, self.dist,
                               (self.rank,) if self.delivers else (), self.stats, transient=self.sink is not None,
                               src_ranks=self.src_ranks, tables_only=self.lean)
            if self.decode is not None:
                got = self.decode(got)
            if self.sink is not None:
                if got:
                    self.sink(got)
                self.delivered += len(got)
            else:
                self.result.update(got)
            self.k += 1
            return
        self._raise()
        t0 = time.perf_counter()
        if self.delivers:
            self._slots.acquire()  # at most two rounds in flight: the page-locked slot k & 1 must have been consumed
            self._raise()
        t1 = time.perf_counter()
        buf = store.pack(n_cap, data_cap)
        tbl = HDR + ENT * n_cap
        _exchange_stats(self.stats, store, buf, self.dist, tbl if self.lean else None)
        work = ev = out = None
        if self.lean:
            out = torch.empty(self.world * tbl, dtype=torch.float32, device=buf.device)
            work = self.dist.all_gather_into_tensor(out, buf[:tbl], async_op=True)
            view = buf.view(1, -1)   # this rank's own rows come out of its own buffer
        elif self.dist is not None:
            out = torch.empty(self.world * buf.numel(), dtype=torch.float32, device=buf.device)
            work = self.dist.all_gather_into_tensor(out, buf, async_op=True)
            view = out.view(self.world, -1)
        else:
            view = buf.view(1, -1)
        if (self.lean or self.dist is None) and self.device.type == "cuda":
            ev = torch.cuda.Event()   # the copy stream must see the pack (a collective's wait() orders only the collective)
            ev.record(torch.cuda.current_stream(self.device))
        self.t["submit_wait_s"] += t1 - t0
        self.t["pack_s"] += time.perf_counter() - t1
        if self.delivers:
            self._q.put((self.k, work, ev, view, n_cap, buf, out))  # buf / out ride along: alive until delivered
        else:
            self._pending.append((work, buf, out))
            while len(self._pending) > 2:
                self._pending.pop(0)[0].wait()
        self.k += 1

    def close(self):
        """Stop the delivery thread (idempotent; rounds already queued are still delivered first).  Callers put this in
        a ``finally`` so that a run that fails half-way does not leave a thread behind."""
        if self._thread is not None:
            self._q.put(None)
            self._thread.join()
            self._thread = None

    def finish(self):
        """Wait for every round to be delivered (and for every collective this rank took part in).  Returns the number
        of waveforms delivered to the sink, or the result dict when there is no sink."""
        self.close()
        for w, _, _ in self._pending:
            if w is not None:
                w.wait()
        self._pending = []
        self._raise()
        if self.stats is not None:
            for k, v in self.t.items():
                self.stats[k] = self.stats.get(k, 0.0) + v
            self.stats["overlap"] = bool(self.overlap)
        return self.delivered if self.sink is not None else self.result


def _exchange_stats(stats, store, buf, dist, sent=None):
    if stats is not None:
        stats["payload_floats"] = stats.get("payload_floats", 0) + sum(int(b[1].sum()) for b in store.batches)
        stats["sent_floats"] = stats.get("sent_floats", 0) + (buf.numel() if sent is None else int(sent))
        stats["collectives"] = stats.get("collectives", 0) + (1 if dist is not None else 0)


def _check_tables(tables, n_cap, stats=None):
    """tables: f32 [world, HDR + ENT * n_cap] (any device), the row tables of a tables-only round.  Validates every rank's
    header and counts what the round delivered over all ranks (``stats['rows_all_ranks']``, ``['floats_all_ranks']``):
    the accounting the full all-gather gave for free."""
    t = (tables.cpu() if tables.is_cuda else tables).contiguous().numpy().view(np.int32).reshape(tables.shape[0], 1 + n_cap, 4)
    rows = used = 0
    for r in range(t.shape[0]):
        n, u = int(t[r, 0, 0]), int(t[r, 0, 2:4].copy().view(np.int64)[0])
        if n < 0 or n > n_cap or u < 0:
            raise ValueError(f"rank {r}: corrupt exchange header (rows {n}, data floats {u})")
        rows, used = rows + n, used + u
    if stats is not None:
        stats["rows_all_ranks"] = stats.get("rows_all_ranks", 0) + rows
        stats["floats_all_ranks"] = stats.get("floats_all_ranks", 0) + used
    return rows, used


def gather_waves(local_waves, local_ids, lengths, parts, hop, rank, world_size, device, dist=None,
                 unpack_ranks=None):
    """List-of-tensors front end of gather_store (geometry from the global job list)."""
    n_cap, data_cap = pack_geometry(lengths, parts, hop)
    st = WaveStore(device)
    for w, j in zip(local_waves, local_ids):
        if w.numel() == 0:
            st.add_empty([j])
        else:
            st.add(w.reshape(1, -1).contiguous(), [w.numel()], [j])
    return gather_store(st, n_cap, data_cap, rank, world_size, dist, unpack_ranks)


def agree_geometry(n_local, data_local, world_size, device, dist=None):
    """(n_cap, data_cap) over ranks when they cannot be derived from the job list (predicted durations
    decide the output lengths): one 16-byte MAX all-reduce ahead of the waveform all-gather."""
    if dist is None:
        return int(n_local), int(data_local)
    t = torch.tensor([int(n_local), int(data_local)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t = t.cpu()
    return int(t[0]), int(t[1])


def _parse_cpulist(txt):
    """'0-15,128-143' -> [[0..15], [128..143]] (one list per contiguous range)"""
    out = []
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.append(list(range(int(a), int(b or a) + 1)))
    return out


def numa_cpus_for(local_rank, gpu_nodes, node_ranges, allowed=None):
    """CPUs for the process that drives GPU ``local_rank``: those of its GPU's NUMA node, split evenly among the local
    ranks whose GPUs sit on the same node (each contiguous CPU range of the node is cut into equal parts, so SMT
    siblings -- the second range of a node on EPYC -- stay with their cores).  ``gpu_nodes[i]``: NUMA node of local GPU
    i (-1 unknown); ``node_ranges[n]``: the node's CPU ranges.  None when nothing can / should be pinned."""
    node = gpu_nodes[local_rank] if 0 <= local_rank < len(gpu_nodes) else -1
    if node < 0 or node not in node_ranges:
        return None
    peers = [i for i, n in enumerate(gpu_nodes) if n == node]
    k, n = peers.index(local_rank), len(peers)
    mine = []
    for rng in node_ranges[node]:
        if allowed is not None:
            rng = [c for c in rng if c in allowed]
        lo, hi = k * len(rng) // n, (k + 1) * len(rng) // n
        mine += rng[lo:hi]
    return sorted(mine) or None


def pin_to_gpu_numa(local_rank, n_local):
    """Pin this process (its Python thread, the Exchange worker, torch's CPU threads) to the CPUs of its GPU's NUMA
    node: page-locked staging buffers are then allocated, filled and written to disk next to the PCIe root the GPU
    hangs off.  ``DISSC_NUMA_PIN=0`` switches it off; any failure to read the topology leaves the affinity alone.
    Returns the CPU list applied, or None."""
    if os.environ.get("DISSC_NUMA_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        nodes = []
        for i in range(max(n_local, local_rank + 1)):
            if i >= torch.cuda.device_count():
                nodes.append(-1)
                continue
            p = torch.cuda.get_device_properties(i)
            bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
                nodes.append(int(f.read().strip()))
        ranges = {}
        for n in set(nodes):
            if n >= 0:
                with open(f"/sys/devices/system/node/node{n}/cpulist") as f:
                    ranges[n] = _parse_cpulist(f.read())
        cpus = numa_cpus_for(local_rank, nodes, ranges, set(os.sched_getaffinity(0)))
        if cpus:
            os.sched_setaffinity(0, cpus)
        return cpus
    except Exception:  # noqa: BLE001  (topology files absent in a container, properties without PCI ids, ...)
        return None


def cgroup_cpu_quota():
    """CPU-time quota of this container in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unreadable."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:  # noqa: BLE001
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:  # noqa: BLE001
        return None


def limit_host_threads(local_world=1, cap=8):
    """The CLIs' host work in torch is the weight-norm fold and a few small tensor ops; torch's default intra-op pool is one
    thread per logical CPU, and STARTING 128 threads under a 16-core CPU quota cost 160 ms of the 210 ms the first fold took
    (tools/first_op_probe.py, profiles/r05/create_timing.txt).  The pool is capped at min(cap, affinity / ranks on the node,
    cgroup quota / ranks); OMP_NUM_THREADS or DISSC_HOST_THREADS (0 = leave torch alone) win.  Returns the thread count set, or
    None when nothing was changed.  Process-wide: for the CLIs (which own their process), not for library users."""
    env = os.environ.get("DISSC_HOST_THREADS")
    if env is not None:
        if int(env) <= 0:
            return None
        torch.set_num_threads(int(env))
        return int(env)
    if os.environ.get("OMP_NUM_THREADS"):
        return None
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = cgroup_cpu_quota()
    if quota is not None:
        avail = min(avail, max(1, int(quota)))
    n = max(1, min(int(cap), avail // max(1, int(local_world))))
    torch.set_num_threads(n)
    return n


def init_distributed(default_port, backend_env="DISSC_DIST_BACKEND"):
    """Process-group set-up shared by the CLIs (sr/inference.py, convert.py): one process per GPU from the
    torchrun environment.  Returns (rank, local_rank, world_size, dist-or-None).

    * WORLD_SIZE > 1: RCCL (backend "nccl") with one GPU per rank; ``DISSC_DIST_BACKEND=gloo`` is the rehearsal
      mode for a box with fewer GPUs than ranks (ranks share devices round-robin, tensors staged through the host).
      Each rank's host threads are pinned to its GPU's NUMA node (``pin_to_gpu_numa``; DISSC_NUMA_PIN=0 = off).
    * WORLD_SIZE == 1: no process group -- unless ``DISSC_FORCE_DIST=1``, which initialises the same backend with
      one rank so that the collectives of the path run on the one GPU that is there.
    Also caps torch's host thread pool (``limit_host_threads``)."""
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    limit_host_threads(int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if world == 1 and os.environ.get("DISSC_FORCE_DIST", "0") != "1":
        return rank, local_rank, world, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(default_port))
    backend = os.environ.get(backend_env, "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    elif torch.cuda.device_count() <= local_rank:
        raise RuntimeError(f"rank {rank}: local rank {local_rank} but {torch.cuda.device_count()} GPUs visible "
                           f"(one GPU per rank; {backend_env}=gloo shares devices for rehearsals)")
    torch.cuda.set_device(local_rank)
    if world > 1 and backend == "nccl":
        pin_to_gpu_numa(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world, dist


OVERLAP_ROUND_FRAMES = 8000  # smallest round worth cutting for overlap: batches under ~4 000 frames lose > 10 %
OVERLAP_MAX_ROUNDS = 4
OVERLAP_MIN_LAST = 4000  # frames per rank the last (smallest) round of the taper must still hold
OVERLAP_CAP_FLOATS = 1 << 26  # output samples per rank and overlapped round: bounds the two page-locked round buffers (256 MB each
                              # per source rank) and the exposed last round of a long run; longer runs simply take more rounds


def overlap_budget(lengths, parts, budget, cap=None):
    """Round budgets (length units per rank and round) when rounds are delivered while the next one computes: a run that
    would fit ONE round is still cut into up to OVERLAP_MAX_ROUNDS rounds of >= OVERLAP_ROUND_FRAMES frames per rank on
    average, so that only the LAST round's exchange + device-to-host copy + sink is exposed -- and the rounds TAPER
    (n : n - 1 : ... : 1 of the largest share), so that the exposed one is the smallest: with n rounds the tail is
    2 / (n (n + 1)) of the run's delivery instead of 1 / n.  ``budget`` (the memory bound of a round, or None) caps every
    entry, and so does ``cap`` (length units per overlapped round, OVERLAP_CAP_FLOATS / hop: a run too long for four rounds
    of that size takes more of them, all but the last of the cap's size).  Derived from the global job list: every rank
    gets the same answer.  Returns ``budget`` itself when the run is too short to cut -- or when the caller already gave
    explicit per-round budgets (a list) --, else a list for plan_rounds."""
    if isinstance(budget, (list, tuple)):
        return budget
    share = max((sum(int(lengths[i]) for i in p) for p in parts), default=0)
    n = min(OVERLAP_MAX_ROUNDS, share // OVERLAP_ROUND_FRAMES)
    while n >= 2 and share // (n * (n + 1) // 2) < OVERLAP_MIN_LAST:
        n -= 1
    if n < 2:
        return budget
    # plan_rounds closes a round when the next job would exceed its budget, so a round holds between budget - longest
    # job and budget: centred on the taper the first n - 1 rounds leave the last one its share give or take
    # (n - 1) / 2 jobs; the last round takes whatever is left (up to the memory bound)
    longest = max((int(v) for v in lengths), default=0)
    tri = n * (n + 1) // 2
    cuts = [-(-share * (n - k) // tri) + longest // 2 for k in range(n - 1)]
    cuts.append(share if budget is None else int(budget))
    lim = min(v for v in (budget, cap, share) if v is not None)
    return [min(int(lim), c) for c in cuts]


def run_resynthesis(generator, jobs, rank=0, world_size=1, device="cuda:0", dist=None, max_batch=128,
                    max_frames=MAX_FRAMES, postprocess=None, unpack_ranks=(0,), sink=None,
                    round_floats=ROUND_FLOATS, stats=None, own_rows=False, overlap=None):
    """jobs: list of dicts {code: int array [T], f0: float array [T], spkr: int}.
    Every rank runs its LPT share in length-bucketed batches, one all-gather per round.  Returns {job_id: float32
    samples} (on ``unpack_ranks``; None = all) -- or, with ``sink``, calls ``sink({job_id: samples})`` once per round
    on those ranks and returns the number of waveforms delivered (nothing is retained between rounds: the arrays a
    sink receives are views of a reused page-locked buffer, valid until it returns).  ``own_rows=True``: every rank
    delivers the rows it decoded itself instead (Exchange).
    Rounds: one unless a rank's share exceeds ``round_floats`` output samples -- or, with a sink on a GPU
    (``overlap``, default on), up to 4 tapering rounds (>= 8 000 frames per rank on average), delivered by a worker
    thread WHILE the next round computes (Exchange; with a sink the callback therefore runs on that thread).
    ``postprocess(wav[B,1,L], n_samples[B])`` runs on the GPU in place.  ``stats`` (a dict) is filled with
    per-rank timing and traffic figures."""
    import time
    lengths = [len(j["code"]) for j in jobs]
    for j, job in enumerate(jobs):
        if len(job["f0"]) != lengths[j]:
            raise ValueError(f"job {j}: {lengths[j]} units but {len(job['f0'])} f0 values")
    parts = lpt_shard(lengths, world_size)
    hop = int(np.prod(generator.h["upsample_rates"]))  # same on every rank, even one with no jobs
    dev = torch.device(device)
    cuda = dev.type == "cuda"
    ex = Exchange(rank, world_size, dev, dist, unpack_ranks, own_rows, sink, stats, overlap=overlap)
    if isinstance(round_floats, (list, tuple)):  # explicit per-round budgets (bench.py's emulation of an N-rank run)
        budget = [max(1, int(f) // hop) for f in round_floats]
    else:
        budget = None if round_floats is None else max(1, round_floats // hop)
    if ex.plan_cut:  # (agreed between the ranks: Exchange)
        budget = overlap_budget(lengths, parts, budget, cap=max(1, OVERLAP_CAP_FLOATS // hop))
    rounds = plan_rounds(lengths, parts, budget)
    t_run = time.perf_counter()
    try:
        return _run_rounds(generator, jobs, rank, dev, cuda, ex, rounds, lengths, parts, hop, max_batch, max_frames,
                           postprocess, stats, t_run)
    finally:
        ex.close()


def _run_rounds(generator, jobs, rank, dev, cuda, ex, rounds, lengths, parts, hop, max_batch, max_frames, postprocess,
                stats, t_run):
    """the round loop of run_resynthesis (its own function so that the caller can close the Exchange in a finally)"""
    import time
    t_host = 0.0
    events = []
    for shares in rounds:
        t0 = time.perf_counter()
        store = WaveStore(dev)
        batches = make_batches(shares[rank], lengths, max_batch, max_frames)
        # ONE upload per round: the padded inputs of every batch go into one host buffer each (page-locked on a GPU) and
        # over with four copies; the batches are views of the device copies.  (A pageable per-batch .to(device) is
        # ordered behind the previous batch's kernels on the stream and blocks the host until they are done: host
        # batching and GPU compute then alternate instead of overlapping.)
        shapes = [(len(bt), max((lengths[i] for i in bt), default=0)) for bt in batches]
        tot = sum(B * T for B, T in shapes)
        nrow = sum(B for B, _ in shapes)
        h_code = torch.zeros(max(tot, 1), dtype=torch.int64, pin_memory=cuda)
        h_f0 = torch.zeros(max(tot, 1), dtype=torch.float32, pin_memory=cuda)
        h_spkr = torch.zeros(max(nrow, 1), dtype=torch.int64, pin_memory=cuda)
        h_lens = torch.zeros(max(nrow, 1), dtype=torch.int32, pin_memory=cuda)
        n_code, n_f0, n_spkr, n_lens = h_code.numpy(), h_f0.numpy(), h_spkr.numpy(), h_lens.numpy()
        o = r = 0
        for bt, (B, T) in zip(batches, shapes):
            for k, i in enumerate(bt):
                n = lengths[i]
                n_code[o + k * T:o + k * T + n] = jobs[i]["code"]
                n_f0[o + k * T:o + k * T + n] = jobs[i]["f0"]
                n_spkr[r + k] = jobs[i]["spkr"]
                n_lens[r + k] = n
            o += B * T
            r += B
        if hasattr(generator, "validate_host_ids"):
            generator.validate_host_ids(h_code[:tot], h_spkr[:nrow])
        if cuda and stats is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream(dev))
        d_code, d_f0 = h_code.to(dev, non_blocking=True), h_f0.to(dev, non_blocking=True)
        d_spkr, d_lens = h_spkr.to(dev, non_blocking=True), h_lens.to(dev, non_blocking=True)
        o = r = 0
        for bt, (B, T) in zip(batches, shapes):
            if T == 0:
                # empty `units` lines: empty waveforms (the generator rejects T = 0).  Never abort one rank
                # here -- the others would wait in the all-gather forever.
                store.add_empty(bt)
                r += B
                continue
            lens = d_lens[r:r + B]
            y = generator(code=d_code[o:o + B * T].view(B, T), f0=d_f0[o:o + B * T].view(B, 1, T),
                          spkr=d_spkr[r:r + B].view(B, 1), lengths=lens)
            assert y.shape[-1] == hop * T
            if postprocess is not None:
                postprocess(y, lens * hop)
            store.add(y, n_lens[r:r + B].astype(np.int64) * hop, bt)
            o += B * T
            r += B
        if cuda and stats is not None:
            ev1.record(torch.cuda.current_stream(dev))
            events.append((ev0, ev1))
        t_host += time.perf_counter() - t0
        n_cap, data_cap = pack_geometry(lengths, shares, hop)
        ex.submit(store, n_cap, data_cap)
        store.clear()
    if stats is not None:
        # this rank's own work, delivery of other ranks' rounds excluded: the GPU spans of its rounds (H2D -> last
        # kernel, HIP events) on a GPU -- launches run ahead of the device, so the host time alone says little -- or
        # the host time of the loop on the CPU
        if cuda:
            torch.cuda.synchronize(dev)
            stats["compute_s"] = sum(a.elapsed_time(b) for a, b in events) * 1e-3
        else:
            stats["compute_s"] = t_host
        stats["host_batching_s"] = t_host
        stats["compute_done_s"] = time.perf_counter() - t_run  # run start -> this rank's last kernel finished
        stats["imbalance"] = imbalance(lengths, parts)
        stats["rounds"] = len(rounds)
    return ex.finish()
