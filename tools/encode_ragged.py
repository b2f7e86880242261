#!/usr/bin/env python
"""HuBERT unit-encode on padded / ragged batches: the same 320 s of audio as 32 x 10 s, as 32 x 10 s inside rows of 12 s,
and with lengths spread over 8-12 s (ms per batch)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dissc_amd.hubert import HubertEncoder
import synthdata as synth

enc = HubertEncoder(synth.synth_hubert_state_dict(6), synth.synth_kmeans_centers(), 6).to("cuda:0")
rs = np.random.RandomState(0)
base = torch.from_numpy(np.stack([synth.synth_waveform(192000, seed=i) for i in range(32)])).cuda()


def timeit(wav, ns):
    for _ in range(2):
        enc(wav, n_samples=ns, want_dense=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        enc(wav, n_samples=ns, want_dense=False)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5


u = torch.full((32,), 160000, dtype=torch.int32)
print(f"32 x 10 s                       : {timeit(base[:, :160000].contiguous(), u):.2f} ms")
for rows in (10.1, 10.5, 11.0, 12.0):
    print(f"32 x 10 s in rows of {rows:4.1f} s     : {timeit(base[:, :int(rows * 16000)].contiguous(), u):.2f} ms")
half = rs.randint(-32000, 32001, size=16)
ns = np.concatenate([160000 + half, 160000 - half]).astype(np.int32)
print(f"lengths {ns.min() / 16000:.1f}-{ns.max() / 16000:.1f} s (320 s in all)  : {timeit(base[:, :int(ns.max())].contiguous(), torch.from_numpy(ns)):.2f} ms")


# two half batches on two streams (two encoder handles: separate workspaces): do they fill each other's tails?
enc2 = HubertEncoder(synth.synth_hubert_state_dict(6), synth.synth_kmeans_centers(), 6).to("cuda:0")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timeit2(wav, ns):
    order = torch.argsort(ns, descending=True)
    ia, ib = order[0::2], order[1::2]          # two halves of equal length mix
    wa, wb = wav[ia.cuda()].contiguous(), wav[ib.cuda()].contiguous()
    na, nb = ns[ia].contiguous(), ns[ib].contiguous()
    def once():
        with torch.cuda.stream(s1):
            enc(wa, n_samples=na, want_dense=False)
        with torch.cuda.stream(s2):
            enc2(wb, n_samples=nb, want_dense=False)
    for _ in range(2):
        once()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(5):
        once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 200


print(f"two streams x 16: 32 x 10 s                 : {timeit2(base[:, :160000].contiguous(), u):.2f} ms")
print(f"two streams x 16: 32 x 10 s in rows of 12 s : {timeit2(base, u):.2f} ms")
nst = torch.from_numpy(ns)
print(f"two streams x 16: lengths 8.2-11.8 s        : {timeit2(base[:, :int(ns.max())].contiguous(), nst):.2f} ms")
