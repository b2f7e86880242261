"""Generate golden vectors by running the REFERENCE itself (this container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Imports the reference modules unmodified from /root/reference (which does not
exist on the GPU box -- only the .npz outputs travel).  Weights/inputs come
from oracle/synth.py (numpy RandomState, reproducible anywhere), are loaded
into the reference modules with strict ``load_state_dict`` (which also pins our
checkpoint key/shape layout against the reference's), and the reference's
outputs are stored.  Nothing of the reference's source is stored.
"""
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import synth  # noqa: E402


def _import_ref_sr():
    sys.path.insert(0, os.path.join(REF, "sr"))
    import models as ref_models  # reference sr/models.py
    import utils as ref_utils  # reference sr/utils.py
    return ref_models, ref_utils


def make_generator():
    ref_models, ref_utils = _import_ref_sr()
    h = ref_utils.AttrDict(json.load(open(os.path.join(REF, "sr/configs/VCTK/hubert100_lut.json"))))
    out = {}
    for seed in (0,):
        g = ref_models.CodeGenerator(h)
        sd = synth.synth_generator_state_dict(seed=seed)
        g.load_state_dict(sd, strict=True)
        g.eval()
        g.remove_weight_norm()
        fsd = g.state_dict()
        # checksums of the folded weights (pins fold_weight_norm incl. ConvTranspose)
        for name in ("conv_pre", "ups.0", "ups.3", "resblocks.0.convs1.2", "resblocks.14.convs2.0", "conv_post"):
            wt = fsd[name + ".weight"].double()
            out[f"s{seed}/fold/{name}"] = np.array([wt.sum().item(), wt.abs().sum().item(), (wt * wt).sum().item()])
        out[f"s{seed}/fold/ups.4.weight"] = fsd["ups.4.weight"].numpy()

        # activations are captured with forward hooks on the reference modules
        def run(code, f0, spkr, want_taps):
            taps = {}
            hooks = []
            if want_taps:
                hooks.append(g.conv_pre.register_forward_hook(lambda m, i, o: taps.__setitem__("conv_pre", o.clone())))
                for i, up in enumerate(g.ups):
                    hooks.append(up.register_forward_hook(lambda m, i_, o, i=i: taps.__setitem__(f"up{i}", o.clone())))
                # the MRF output of stage i is the (pre-lrelu) input of ups[i+1] / conv_post
                for i, up in enumerate(g.ups):
                    if i > 0:
                        hooks.append(up.register_forward_pre_hook(
                            lambda m, a, i=i: taps.__setitem__(f"lrelu_mrf{i-1}", a[0].clone())))
                for j, rb in enumerate(g.resblocks):
                    hooks.append(rb.register_forward_hook(lambda m, i_, o, j=j: taps.__setitem__(f"rb{j}", o.clone())))
            with torch.no_grad():
                y = g(code=torch.from_numpy(code), f0=torch.from_numpy(f0), spkr=torch.from_numpy(spkr))
            for hk in hooks:
                hk.remove()
            return y, taps

        for T in (1, 2, 7, 33, 99):
            code, f0, spkr, _ = synth.synth_generator_inputs(1, T, seed=100 + T)
            y, taps = run(code, f0, spkr, want_taps=(T == 7))
            out[f"s{seed}/T{T}/wav"] = y.numpy()
            for k, v in taps.items():
                out[f"s{seed}/T{T}/{k}"] = v.numpy()
        # ragged batch: the reference never batches -> one call per utterance
        code, f0, spkr, _ = synth.synth_generator_inputs(4, 40, seed=777)
        lengths = np.array([40, 23, 9, 1], dtype=np.int32)
        for b in range(4):
            n = int(lengths[b])
            y, _ = run(code[b:b + 1, :n], f0[b:b + 1, :, :n], spkr[b:b + 1], False)
            out[f"s{seed}/ragged/wav{b}"] = y.numpy()
        out[f"s{seed}/ragged/lengths"] = lengths
    np.savez_compressed(os.path.join(HERE, "gen_vctk.npz"), **out)
    print("gen_vctk.npz", len(out), "arrays")


if __name__ == "__main__":
    which = sys.argv[1:] or ["generator"]
    if "generator" in which:
        make_generator()
