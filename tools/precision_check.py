"""Waveform error of the current arithmetic mode (DISSC_OPTIONS=precision=0|1) vs the reference goldens."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dissc_amd, synthdata as synth
g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to("cuda:0")
g.load_state_dict(synth.synth_generator_state_dict(0)); g.eval().remove_weight_norm()
gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'gen_vctk.npz'))
for T in (1, 7, 33, 99):
    code, f0, spkr, _ = synth.synth_generator_inputs(1, T, seed=100 + T)
    y = g(code=torch.from_numpy(code), f0=torch.from_numpy(f0), spkr=torch.from_numpy(spkr)).cpu().numpy()
    ref = gold[f"s0/T{T}/wav"]
    e = y - ref
    print(T, "rms_err %.3e max %.3e ref_rms %.3f" % (np.sqrt((e**2).mean()), np.abs(e).max(), np.sqrt((ref**2).mean())))
