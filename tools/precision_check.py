"""Waveform error of the current arithmetic mode (DISSC_OPTIONS=precision=0|1) vs the reference
goldens: each golden utterance alone (B=1), then all of them as ONE ragged batch, which must
reproduce the B=1 waveforms bit for bit (utterances are independent in every kernel)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dissc_amd  # noqa: E402
import synthdata as synth  # noqa: E402

g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to("cuda:0")
g.load_state_dict(synth.synth_generator_state_dict(0))
g.eval().remove_weight_norm()
gold = np.load(os.path.join(ROOT, "tests", "golden", "gen_vctk.npz"))
LENS = (1, 7, 33, 99)
single, inputs = {}, {}
for T in LENS:
    code, f0, spkr, _ = synth.synth_generator_inputs(1, T, seed=100 + T)
    inputs[T] = (code, f0, spkr)
    y = g(code=torch.from_numpy(code), f0=torch.from_numpy(f0), spkr=torch.from_numpy(spkr)).cpu().numpy()
    single[T] = y
    ref = gold[f"s0/T{T}/wav"]
    e = y - ref
    print(T, "rms_err %.3e max %.3e ref_rms %.3f" % (np.sqrt((e ** 2).mean()), np.abs(e).max(),
                                                      np.sqrt((ref ** 2).mean())))

Tm = max(LENS)
code = np.zeros((len(LENS), Tm), dtype=np.int64)
f0 = np.zeros((len(LENS), 1, Tm), dtype=np.float32)
spkr = np.zeros((len(LENS), 1), dtype=np.int64)
for b, T in enumerate(LENS):
    code[b, :T], f0[b, 0, :T], spkr[b, 0] = inputs[T][0][0], inputs[T][1][0, 0], inputs[T][2][0, 0]
y = g(code=torch.from_numpy(code), f0=torch.from_numpy(f0), spkr=torch.from_numpy(spkr),
      lengths=torch.tensor(LENS, dtype=torch.int32)).cpu().numpy()
hop = y.shape[-1] // Tm
for b, T in enumerate(LENS):
    same = np.array_equal(y[b, :, :T * hop], single[T][0])
    tail = float(np.abs(y[b, :, T * hop:]).max()) if T < Tm else 0.0
    print("ragged", T, "bitwise_equal_to_single", int(same), "tail_max %.1e" % tail)
