"""Maximum-size probe: tensors beyond 2^31 elements / 2^32 bytes per activation buffer (index arithmetic), checked by
batch independence -- rows of the giant batch must be bit-identical to the same utterance run on its own.
    python tools/max_size_probe.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthdata as synth  # noqa: E402
import dissc_amd  # noqa: E402
from dissc_amd.hubert import HubertEncoder  # noqa: E402

dev = torch.device("cuda:0")


def make_generator():
    g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to(dev)
    g.load_state_dict(synth.synth_generator_state_dict(seed=0))
    g.eval().remove_weight_norm()
    return g


def gen_case(g, B, T, rows, ragged=False):
    code, f0, spkr, _ = synth.synth_generator_inputs(B, T, seed=5)
    lengths = None
    if ragged:
        lengths = torch.from_numpy(np.random.RandomState(1).randint(1, T + 1, B).astype(np.int32))
        lengths[rows[0]] = T
    tc, tf, ts = torch.from_numpy(code), torch.from_numpy(f0), torch.from_numpy(spkr)
    torch.cuda.synchronize()
    t = time.perf_counter()
    kw = {} if lengths is None else {"lengths": lengths}
    y = g(code=tc, f0=tf, spkr=ts, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    ok = bool(torch.isfinite(y).all())
    for b in rows:
        n = T if lengths is None else int(lengths[b])
        y1 = g(code=tc[b:b + 1, :n], f0=tf[b:b + 1, :, :n], spkr=ts[b:b + 1])
        same = torch.equal(y1[0, 0], y[b, 0, :n * 320])
        tail = (not bool(y[b, 0, n * 320:].any()))
        ok = ok and same and tail
        if not (same and tail):
            d = (y1[0, 0] - y[b, 0, :n * 320]).abs()
            print(f"   row {b}: MISMATCH max {float(d.max()):.3e} at {int(d.argmax())} tail_zero={tail}")
    print(f"generator B={B} T={T} ragged={ragged}: elements/buffer {B * T * 320 * 16:.3e}, {dt * 1e3:.0f} ms, ok={ok}", flush=True)
    del y
    torch.cuda.empty_cache()
    return ok


def hubert_case(enc, B, sec, rows=None):
    n = int(sec * 16000)
    wav = torch.stack([torch.from_numpy(synth.synth_waveform(n, seed=100 + (b % 7))) for b in range(B)]).to(dev)
    torch.cuda.synchronize()
    t = time.perf_counter()
    e = enc(wav, want_dense=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    good = True
    for b in rows or (0, 6, B // 2, B - 1):
        e1 = enc(wav[b:b + 1], want_dense=False)
        same = torch.equal(e1["units"][0], e["units"][b])
        good &= same
        if not same:
            print(f"   row {b}: {int((e1['units'][0] != e['units'][b]).sum())} units differ")
    print(f"hubert B={B} x {sec:g} s: conv0 elements {512 * (n // 5) * B:.3e}, {dt * 1e3:.0f} ms, ok={good}", flush=True)
    return good


def make_encoder():
    return HubertEncoder(synth.synth_hubert_state_dict(), torch.as_tensor(synth.synth_kmeans_centers()), 6).to(dev)


def main():
    g = make_generator()
    ok = True
    ok &= gen_case(g, 1700, 250, [0, 1, 849, 1342, 1343, 1699])           # > 2^31 elements per activation buffer
    ok &= gen_case(g, 1700, 250, [0, 3, 849, 1342, 1343, 1699], True)
    ok &= gen_case(g, 3, 20000, [0, 2])                                    # 400 s utterances
    ok &= gen_case(g, 900, 500, [0, 450, 671, 672, 899])                   # 2.3e9 elements
    del g
    enc = make_encoder()
    ok &= hubert_case(enc, 170, 10.0)
    ok &= hubert_case(enc, 40, 60.0)
    print("ALL OK" if ok else "FAILURES")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
