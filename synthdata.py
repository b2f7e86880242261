"""Deterministic synthetic checkpoints and inputs for tests, benchmarks and profiling.

Not part of the oracle and not part of the compute path: pure numpy/torch generators of
random-but-reproducible weights in the reference's checkpoint layouts and of input batches.

No real DISSC checkpoint exists offline (Google-Drive links, reference
README.md:76,93,117,142), so parity is pinned on synthetic weights laid out
exactly like the reference's checkpoints:

* vocoder  ``g_########`` = ``{'generator': state_dict}`` with 97 weight-normed
  conv layers x {bias, weight_g, weight_v} + ``dict.weight`` + ``spkr.weight``
  (reference sr/train.py:205-214, sr/models.py:72-96,125-135).
* predictors ``best_model.pth`` = plain ``state_dict`` (reference
  train_len_predictor.py:101-103, train_f0_predictor.py:98-100).

Everything is drawn from ``numpy.random.RandomState`` (frozen legacy stream) so
the same seed gives the same bytes in this container and on the GPU box.
"""
from collections import OrderedDict

import numpy as np
import torch

VCTK_CONFIG = {
    "resblock": "1",
    "upsample_rates": [5, 4, 4, 2, 2],
    "upsample_kernel_sizes": [11, 8, 8, 4, 4],
    "upsample_initial_channel": 512,
    "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    "num_embeddings": 100,
    "embedding_dim": 128,
    "model_in_dim": 257,
    "code_hop_size": 320,
    "f0": True,
    "multispkr": "_",
    "f0_normalize": False,
    "sampling_rate": 16000,
}


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a.astype(np.float32)))


def _wn_conv(rs, sd, name, shape, fan_in, gain, g_dim0):
    """weight_v ~ N(0,1); weight_g = ||v|| * N(1,0.1)*gain/sqrt(fan_in)*sqrt(numel/g_dim0)
    so the folded weight has std ~ gain/sqrt(fan_in)."""
    v = rs.standard_normal(shape)
    norm = np.sqrt((v.reshape(shape[0], -1) ** 2).sum(1))
    per = np.sqrt(np.prod(shape[1:]))
    g = (gain / np.sqrt(fan_in)) * per * (1.0 + 0.1 * rs.standard_normal(shape[0]))
    # g multiplies v/||v||: effective per-element std = g/per
    del norm
    sd[name + ".weight_g"] = _t(g.reshape((shape[0],) + (1,) * (len(shape) - 1)))
    sd[name + ".weight_v"] = _t(v)


def synth_generator_state_dict(h=None, seed=0):
    """State dict with the reference's 293 keys."""
    h = h or VCTK_CONFIG
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    c0 = h["upsample_initial_channel"]
    in_dim = h.get("model_in_dim", 128)
    # conv_pre: Conv1d(in_dim, c0, 7)
    sd["conv_pre.bias"] = _t(0.1 * rs.standard_normal(c0))
    _wn_conv(rs, sd, "conv_pre", (c0, in_dim, 7), in_dim * 7, 1.0, c0)
    # ups: ConvTranspose1d weight [Cin, Cout, k]; weight_g is per *input* channel
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        cin, cout = c0 // 2 ** i, c0 // 2 ** (i + 1)
        sd[f"ups.{i}.bias"] = _t(0.1 * rs.standard_normal(cout))
        _wn_conv(rs, sd, f"ups.{i}", (cin, cout, k), cin * k / u, 1.4, cin)
    # resblocks
    nk = len(h["resblock_kernel_sizes"])
    for i in range(len(h["upsample_rates"])):
        ch = c0 // 2 ** (i + 1)
        for j, k in enumerate(h["resblock_kernel_sizes"]):
            idx = i * nk + j
            for grp, gain in (("convs1", 1.4), ("convs2", 0.6)):
                for m in range(3):
                    name = f"resblocks.{idx}.{grp}.{m}"
                    sd[name + ".bias"] = _t(0.1 * rs.standard_normal(ch))
                    _wn_conv(rs, sd, name, (ch, ch, k), ch * k, gain, ch)
    ch = c0 // 2 ** len(h["upsample_rates"])
    sd["conv_post.bias"] = _t(0.05 * rs.standard_normal(1))
    _wn_conv(rs, sd, "conv_post", (1, ch, 7), ch * 7, 0.35, 1)
    sd["dict.weight"] = _t(rs.standard_normal((h["num_embeddings"], h["embedding_dim"])))
    sd["spkr.weight"] = _t(rs.standard_normal((200, h["embedding_dim"])))
    return sd


def synth_generator_inputs(B, T, seed=1234, ragged=False, n_spk=108, n_codes=100):
    """runs of a uniform symbol (geometric, mean 2.5 frames),
    f0 ~ N(0,1) with ~35% exact-zero unvoiced runs, spkr uniform."""
    rs = np.random.RandomState(seed)
    code = np.zeros((B, T), dtype=np.int64)
    f0 = np.zeros((B, 1, T), dtype=np.float32)
    for b in range(B):
        t = 0
        while t < T:
            run = rs.geometric(1 / 2.5)
            code[b, t:t + run] = rs.randint(0, n_codes)
            t += run
        f0[b, 0] = rs.standard_normal(T)
        t = 0
        while t < T:
            run = rs.geometric(1 / 12.0)
            if rs.rand() < 0.35:
                f0[b, 0, t:t + run] = 0.0
            t += run
    spkr = rs.randint(0, n_spk, size=(B, 1)).astype(np.int64)
    if ragged:
        lengths = rs.randint(max(1, T // 2), T + 1, size=B).astype(np.int32)
        lengths[0] = T
    else:
        lengths = np.full(B, T, dtype=np.int32)
    return code, f0, spkr, lengths


# ----------------------------------------------------------------------------------------------
# predictors (reference model/len_predictor.py, model/pitch_predictor.py)
# ----------------------------------------------------------------------------------------------
def _conv(rs, sd, name, cout, cin, k, gain=1.4):
    sd[name + ".weight"] = _t(rs.standard_normal((cout, cin, k)) * gain / np.sqrt(cin * k))
    sd[name + ".bias"] = _t(0.1 * rs.standard_normal(cout))


def _bn(rs, sd, name, c=128):
    sd[name + ".weight"] = _t(1.0 + 0.2 * rs.standard_normal(c))
    sd[name + ".bias"] = _t(0.1 * rs.standard_normal(c))
    sd[name + ".running_mean"] = _t(0.2 * rs.standard_normal(c))
    sd[name + ".running_var"] = _t(0.5 + rs.rand(c))
    sd[name + ".num_batches_tracked"] = torch.tensor(1000, dtype=torch.int64)


def synth_len_state_dict(n_tokens=100, n_speakers=108, seed=1):
    """53 keys: token_emb, spk_emb, cnn1/bn1, cnn11..16/bn11..16, cnn2."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    sd["token_emb.weight"] = _t(rs.standard_normal((n_tokens + 1, 32)))
    sd["spk_emb.weight"] = _t(rs.standard_normal((n_speakers, 32)))
    _conv(rs, sd, "cnn1", 128, 64, 3)
    _bn(rs, sd, "bn1")
    for i in range(1, 7):
        _conv(rs, sd, f"cnn1{i}", 128, 128, 3)
        _bn(rs, sd, f"bn1{i}")
    _conv(rs, sd, "cnn2", 1, 128, 3, gain=0.12)  # lens ~ 2.6 +- 0.5 for any seed / speaker count
    return sd


def synth_len_norm_stats():
    """len_norm_stats.pth = (mean, std) tensors (reference train_len_predictor.py:32)."""
    return torch.tensor(2.6), torch.tensor(1.7)


def synth_pitch_state_dict(kind="new", n_tokens=100, n_speakers=108, seed=2):
    """'new': 34 keys incl. buffer pe.pe [1,850,32]; 'base': 78 keys (BN after every conv but cnn2)."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    sd["token_emb.weight"] = _t(rs.standard_normal((n_tokens + 1, 32)))
    sd["spk_emb.weight"] = _t(rs.standard_normal((n_speakers + 1, 32)))
    if kind == "new":
        lin = torch.linspace(0, 1, 850).unsqueeze(-1)
        sd["pe.pe"] = torch.cat([lin.repeat_interleave(16, -1), torch.linspace(1, 0, 850).unsqueeze(-1)
                                 .repeat_interleave(16, -1)], -1).unsqueeze(0)
    names = ["cnn1"] + [f"cnn1{i}" for i in range(1, 8)]
    for n in names:
        _conv(rs, sd, n, 128, 64 if n == "cnn1" else 128, 3)
        if kind == "base":
            _bn(rs, sd, "bn" + n[3:])
    if kind == "new":
        _bn(rs, sd, "bn2")
    _conv(rs, sd, "cnn2", 128, 128, 3)
    _conv(rs, sd, "cnn_class1", 128, 128, 3)
    if kind == "base":
        _bn(rs, sd, "bn_c1")
    _conv(rs, sd, "cnn_class2", 1, 128, 1, gain=1.0)
    _conv(rs, sd, "cnn_reg1", 128, 128, 3)
    if kind == "base":
        _bn(rs, sd, "bn_r1")
    _conv(rs, sd, "cnn_reg2", 1, 128, 1, gain=1.0)
    return sd


def synth_unit_sequences(n, T_lo=60, T_hi=500, seed=99, n_codes=100):
    """Random unit sequences with geometric run lengths (mean 2.5), like encode output."""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        T = int(rs.randint(T_lo, T_hi + 1))
        seq = np.zeros(T, dtype=np.int64)
        t = 0
        prev = -1
        while t < T:
            run = rs.geometric(1 / 2.5)
            c = int(rs.randint(0, n_codes))
            if c == prev:
                c = (c + 1) % n_codes
            seq[t:t + run] = c
            prev = c
            t += run
        out.append(seq)
    return out


# ----------------------------------------------------------------------------------------------
# HuBERT-base (fairseq checkpoint key names) + k-means centroids
# ----------------------------------------------------------------------------------------------
def synth_hubert_state_dict(n_layers=6, seed=3):
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    convs = [(512, 1, 10)] + [(512, 512, 3)] * 4 + [(512, 512, 2)] * 2
    for i, (co, ci, k) in enumerate(convs):
        sd[f"feature_extractor.conv_layers.{i}.0.weight"] = _t(rs.standard_normal((co, ci, k)) * 1.6 / np.sqrt(ci * k))
    sd["feature_extractor.conv_layers.0.2.weight"] = _t(1.0 + 0.2 * rs.standard_normal(512))
    sd["feature_extractor.conv_layers.0.2.bias"] = _t(0.1 * rs.standard_normal(512))

    def ln(name, c):
        sd[name + ".weight"] = _t(1.0 + 0.1 * rs.standard_normal(c))
        sd[name + ".bias"] = _t(0.05 * rs.standard_normal(c))

    def lin(name, co, ci, gain=1.0):
        sd[name + ".weight"] = _t(rs.standard_normal((co, ci)) * gain / np.sqrt(ci))
        sd[name + ".bias"] = _t(0.05 * rs.standard_normal(co))

    ln("layer_norm", 512)
    lin("post_extract_proj", 768, 512)
    v = rs.standard_normal((768, 48, 128))
    sd["encoder.pos_conv.0.weight_v"] = _t(v)
    sd["encoder.pos_conv.0.weight_g"] = _t((np.sqrt((v ** 2).sum((0, 1), keepdims=True))
                                            * 1.2 / np.sqrt(48 * 128)) * (1 + 0.1 * rs.standard_normal((1, 1, 128))))
    sd["encoder.pos_conv.0.bias"] = _t(0.05 * rs.standard_normal(768))
    ln("encoder.layer_norm", 768)
    for i in range(n_layers):
        p = f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin(p + "self_attn." + n, 768, 768, 1.5 if n in ("q_proj", "k_proj") else 1.0)
        ln(p + "self_attn_layer_norm", 768)
        lin(p + "fc1", 3072, 768)
        lin(p + "fc2", 768, 3072)
        ln(p + "final_layer_norm", 768)
    return sd


def synth_kmeans_centers(k=100, dim=768, seed=4):
    return _t(np.random.RandomState(seed).standard_normal((k, dim)) * 0.7)


def synth_waveform(n, seed=0):
    """N(0,0.1) noise + a few tones, clipped to [-1,1]"""
    rs = np.random.RandomState(seed)
    t = np.arange(n) / 16000.0
    x = 0.1 * rs.standard_normal(n) + 0.2 * np.sin(2 * np.pi * (110 + 40 * rs.rand()) * t) \
        + 0.1 * np.sin(2 * np.pi * (900 + 300 * rs.rand()) * t)
    return np.clip(x, -1, 1).astype(np.float32)
