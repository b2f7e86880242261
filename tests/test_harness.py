"""Host logic of the data-parallel resynthesis harness, incl. the N>1 path on CPU/gloo."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from dissc_amd import harness

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lpt_shard_is_balanced_and_deterministic():
    rs = np.random.RandomState(0)
    lengths = rs.randint(100, 500, size=257).tolist()
    parts = harness.lpt_shard(lengths, 8)
    assert sorted(i for p in parts for i in p) == list(range(257))
    loads = [sum(lengths[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(lengths)
    assert parts == harness.lpt_shard(lengths, 8)
    assert harness.lpt_shard([], 4) == [[], [], [], []]
    assert harness.lpt_shard([5], 2) == [[0], []]


def test_make_batches_limits():
    lengths = [500] * 40 + [100] * 70 + [7]
    ids = list(range(len(lengths)))
    b = harness.make_batches(ids, lengths, max_batch=32, max_frames=32 * 500)
    assert sorted(i for x in b for i in x) == ids
    for x in b:
        assert len(x) <= 32
        assert max(lengths[i] for i in x) * len(x) <= 32 * 500
        assert [lengths[i] for i in x] == sorted((lengths[i] for i in x), reverse=True)


def test_pack_unpack_roundtrip():
    waves = [torch.arange(5, dtype=torch.float32), torch.zeros(0), torch.full((9,), -2.5)]
    buf = harness.pack_waves(waves, [7, 3, 100000], n_max=4, l_max=9, device="cpu")
    assert buf.shape == (4, harness.HDR + 12)  # rows padded to a multiple of 4 floats
    out = harness.unpack_waves(buf)
    assert sorted(out) == [3, 7, 100000]
    np.testing.assert_array_equal(out[7], np.arange(5, dtype=np.float32))
    assert out[3].shape == (0,)
    np.testing.assert_array_equal(out[100000], np.full(9, -2.5, np.float32))


class _FakeGenerator:
    """CPU stand-in with the CodeGenerator call signature: wav = f(code, f0, spkr) per sample,
    so the distributed plumbing can be checked without a GPU."""
    h = {"upsample_rates": [2, 2]}

    def __call__(self, code, f0, spkr, lengths):
        B, T = code.shape
        y = torch.zeros(B, 1, 4 * T)
        for b in range(B):
            n = int(lengths[b])
            v = code[b, :n].float() * 0.001 + f0[b, 0, :n] + spkr[b, 0].float()
            y[b, 0, :4 * n] = v.repeat_interleave(4)
        return y


def _jobs(n=23, seed=3):
    rs = np.random.RandomState(seed)
    jobs = []
    for _ in range(n):
        T = int(rs.randint(1, 40))
        jobs.append(dict(code=rs.randint(0, 100, T), f0=rs.standard_normal(T).astype(np.float32),
                         spkr=int(rs.randint(0, 9))))
    for _ in range(2):  # empty `units` lines must come back as empty waveforms, not abort a rank
        jobs.append(dict(code=np.zeros(0, np.int64), f0=np.zeros(0, np.float32), spkr=1))
    return jobs


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = harness.run_resynthesis(_FakeGenerator(), _jobs(), rank, world, "cpu", dist, max_batch=4,
                                  max_frames=100, unpack_ranks=None)
    only0 = harness.run_resynthesis(_FakeGenerator(), _jobs(), rank, world, "cpu", dist, max_batch=4,
                                    max_frames=100)  # default: only rank 0 unpacks
    assert (len(only0) == 25) == (rank == 0)
    q.put((rank, {k: v.tolist() for k, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
    single = harness.run_resynthesis(_FakeGenerator(), _jobs(), 0, 1, "cpu", None, max_batch=4,
                                     max_frames=100)
    assert sorted(single) == list(range(25)) and single[23].shape == (0,) and single[24].shape == (0,)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(2):  # every rank holds every waveform after the one all-gather
        assert sorted(got[r]) == list(range(25))
        for j in range(25):
            np.testing.assert_array_equal(np.array(got[r][j], dtype=np.float32), single[j])
