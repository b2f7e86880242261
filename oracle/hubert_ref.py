"""CPU restatement of the HuBERT-base unit encoder (TEST INFRASTRUCTURE).

PARITY UNPINNED against the reference's actual dependency: the arithmetic lives in
fairseq (pinned at commit dd106d9534b22e7db859a6b87ffd7780c38341f8, reference README.md:34) and
textlesslib (unpinned HEAD, README.md:31-33), reached from data/encode.py:21-22,32 -- neither
library nor the hubert-base-ls960 / k-means-100 weights exist offline.  What IS pinned
(tests/golden/hubert.npz, made by tests/golden/make_golden.py):
  * this restatement == HuggingFace ``transformers.HubertModel`` (architecture-equivalent to
    fairseq HuBERT-base per its conversion script; hidden_states[6] == extract_features(
    output_layer=6)) on seeded random weights,
  * kmeans_assign == sklearn.cluster.KMeans.predict with the same centroids.

Published algorithm restated (fairseq hubert.py / wav2vec2.py, HuBERT-base config):
  conv_feature_extractor  7 x Conv1d(no bias) k=(10,3,3,3,3,2,2) s=(5,2,2,2,2,2,2), 512 ch;
                          GroupNorm(512 groups) after conv 0; exact GELU after every conv
  project                 LayerNorm(512) -> Linear(512,768)
  encoder                 x += GELU(Conv1d(768,768,k=128,pad=64,groups=16, weight_norm dim=2)[..., :-1]);
                          LayerNorm; 6 x post-LN block {MHA 12x64 (q scaled by 1/8), FFN 768-3072-768 GELU}
  kmeans_assign           argmin_k ||x - c_k||^2 (lowest index on ties)

Weight names follow the fairseq checkpoint (``model`` dict of hubert_base_ls960.pt).

What textlesslib does around the model [3P-unverified: written down from memory of
textless/data/{speech_encoder,hubert_feature_reader,kmeans_quantizer}.py, source absent]:
  * HubertFeatureReader: loads the fairseq checkpoint, ``model.eval()``; the waveform is NOT
    normalised for hubert-base-ls960 (``task.cfg.normalize == False``; only *_large models
    layer-norm the input); inputs longer than ``max_chunk = 1 600 000`` samples are cut into
    chunks that are encoded independently with
    ``extract_features(source=chunk, padding_mask=None, mask=False, output_layer=6)`` and the
    chunk features concatenated along time  -> dissc_amd.hubert.HubertEncoder.MAX_CHUNK;
  * KMeansQuantizer.forward: ``kmeans_model.predict(dense.cpu().numpy())`` on a joblib-pickled
    sklearn ``MiniBatchKMeans(n_clusters=100)``; dense is float32, so sklearn works in float32
    (``cluster_centers_`` are cast to the dtype of X).  sklearn's dense predict
    (``_lloyd_iter_chunked_dense``, chunks of 256 rows) evaluates, per row x and centre c_k,
        d_k = ||c_k||^2 + (-2) * <x, c_k>           (one sgemm with alpha=-2, beta=1 onto a matrix
                                                     pre-filled with the squared centre norms;
                                                     ||x||^2 is omitted: constant per row)
    and takes the FIRST minimum (strict ``<`` while scanning k = 0..K-1).  ``kmeans_assign`` below
    and the HIP ``kmeans_argmin_kernel`` use exactly this expression, order and tie rule; the
    squared centre norms are fp32 sums (summation order inside sklearn's einsum / our loop differs
    by <= 1 ulp-level noise, far below any margin that is not itself a tie);
  * durations: with ``deduplicate=False`` every unit has duration 1; ``f0``: YAAPT, see
    oracle/yaapt_ref.py.
"""
import torch
import torch.nn.functional as F

CONV_LAYERS = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2
EPS = 1e-5


def num_frames(n_samples):
    n = n_samples
    for _, k, s in CONV_LAYERS:
        n = (n - k) // s + 1 if n >= k else 0
    return n


def conv_feature_extractor(sd, wav):
    """wav f32 [B,N] -> [B,512,T]"""
    x = wav.unsqueeze(1)
    for i, (_, k, s) in enumerate(CONV_LAYERS):
        x = F.conv1d(x, sd[f"feature_extractor.conv_layers.{i}.0.weight"], None, stride=s)
        if i == 0:
            x = F.group_norm(x, x.shape[1], sd["feature_extractor.conv_layers.0.2.weight"],
                             sd["feature_extractor.conv_layers.0.2.bias"], EPS)
        x = F.gelu(x)
    return x


def pos_conv_weight(sd):
    # weight_norm(dim=2): norm over dims (0,1) for every kernel position
    return torch._weight_norm(sd["encoder.pos_conv.0.weight_v"], sd["encoder.pos_conv.0.weight_g"], 2)


def encoder(sd, feats, n_layers=6, n_heads=12):
    """feats [B,512,T] -> layer-``n_layers`` output [B,T,768]"""
    x = feats.transpose(1, 2)
    x = F.layer_norm(x, (x.shape[-1],), sd["layer_norm.weight"], sd["layer_norm.bias"], EPS)
    x = F.linear(x, sd["post_extract_proj.weight"], sd["post_extract_proj.bias"])
    pc = F.conv1d(x.transpose(1, 2), pos_conv_weight(sd), sd["encoder.pos_conv.0.bias"], padding=64, groups=16)
    pc = F.gelu(pc[:, :, :-1])
    x = x + pc.transpose(1, 2)
    x = F.layer_norm(x, (768,), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], EPS)
    B, T, D = x.shape
    hd = D // n_heads
    for i in range(n_layers):
        p = f"encoder.layers.{i}."
        q = F.linear(x, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]) * hd ** -0.5
        k = F.linear(x, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
        v = F.linear(x, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        q, k, v = (t.view(B, T, n_heads, hd).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, T, D)
        a = F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        x = F.layer_norm(x + a, (D,), sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"], EPS)
        h = F.gelu(F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
        h = F.linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"])
        x = F.layer_norm(x + h, (D,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], EPS)
    return x


def kmeans_assign(x, centers):
    """x [T,D], centers [K,D] -> int64 [T]: first minimum over k of ||c_k||^2 - 2 <x, c_k>, fp32
    (the expression sklearn's dense predict evaluates; see the module docstring)."""
    d = (centers * centers).sum(1)[None, :] - 2.0 * (x @ centers.t())
    return torch.argmin(d, dim=1)  # torch.argmin returns the first index of the minimum


def kmeans_margin(x, centers):
    """per frame: gap between the best and the second-best value of the expression above (what decides
    whether an fp32-level difference in x may legitimately flip the unit)"""
    d = (centers * centers).sum(1)[None, :] - 2.0 * (x @ centers.t())
    top2 = torch.topk(d, 2, dim=1, largest=False).values
    return top2[:, 1] - top2[:, 0]


FEAT_EPS_L2_REL = 2e-5  # asserted bound on ||x_gpu[t] - x_ref[t]||_2 / ||x_ref[t]||_2, per frame: <= 10x the measured 1.6e-6 ... 3.7e-6
                        # (profiles/r05/s2tc_encode_ab.txt) -- a regression guard, not the acceptance bar (round 5 verdict, weak #1)


def unit_flip_allowed(x_ref, centers, eps_l2=None, eps_rel=FEAT_EPS_L2_REL):
    """Which unit changes a feature error can explain -- DERIVED, not a hand-set margin.

    With s_k(x) = ||c_k||^2 - 2 <x, c_k> (the expression of kmeans_assign) and a feature error delta,
    (s_j - s_i)(x + delta) = (s_j - s_i)(x) + 2 <delta, c_i - c_j>, so a frame whose reference unit is i can come out as j
    only if   s_j(x) - s_i(x) <= 2 ||c_i - c_j||_2 ||delta||_2 + r_ij,
    where r_ij bounds the rounding of the two fp32 evaluations of s (reference and device, each a length-D dot product:
    |fl(s_k) - s_k| <= gamma_D (||c_k||^2 + 2 sum_d |x_d c_kd|), gamma_D = D u / (1 - D u), u = 2^-24).
    ``eps_l2`` [T]: the per-frame ||delta||_2 (measured against the device's features when they are at hand); None =
    the asserted bound ``eps_rel`` * ||x_ref[t]||_2.
    -> (allowed bool [T,K]: allowed[t, j] = "unit j is explicable at frame t" (always True at the reference unit),
        n_ambiguous int: frames with more than one allowed unit)."""
    x = torch.as_tensor(x_ref).double()
    c = torch.as_tensor(centers).double()
    T, D = x.shape
    s = (c * c).sum(1)[None, :] - 2.0 * (x @ c.t())                      # [T,K]
    i = s.argmin(1)
    gap = s - s.gather(1, i[:, None])                                    # s_j - s_i >= 0
    if eps_l2 is None:
        eps_l2 = eps_rel * x.norm(dim=1)
    eps_l2 = torch.as_tensor(eps_l2).double().reshape(T)
    cdist = torch.cdist(c, c)                                            # [K,K]
    u = 2.0 ** -24
    # Rounding of a length-D fp32 dot product.  The worst-case constant gamma_D = D u is ~10x what any summation order
    # delivers on real data (errors of alternating sign: they grow like a random walk) and, at scores of O(1e3), made r
    # dominate the feature term -- a weaker check presented as a stricter one.  Model: the random walk's scale,
    # sqrt(D) u (never above the worst case) -- on the golden features the measured fp32-vs-fp64 score difference is 1.7 u
    # at worst, 16x inside it (tests/test_oracle_golden.py::test_score_rounding_model_holds asserts 4x).
    gamma = min(D * u / (1 - D * u), (D ** 0.5) * u)
    mag = (c * c).sum(1)[None, :] + 2.0 * (x.abs() @ c.abs().t())        # [T,K]
    r = 2.0 * gamma * (mag + mag.gather(1, i[:, None]))                  # both evaluations, both terms
    allowed = gap <= 2.0 * cdist[i] * eps_l2[:, None] + r
    return allowed.numpy(), int((allowed.sum(1) > 1).sum())


def check_units(units, units_ref, x_ref, centers, x_dev=None, tag="units", max_mismatch=1):
    """Assert: every frame where ``units`` differs from the reference's is explained by the feature error (measured when
    ``x_dev`` is given, else the asserted FEAT_EPS_L2_REL bound) -- and the unit chosen instead is one of the explicable
    ones -- and no more than ``max_mismatch`` frames of the utterance differ at all (measured on every golden: 0; None: unbounded,
    for constructed ties).
    Returns (n_mismatch, n_ambiguous)."""
    import numpy as np
    x_ref = torch.as_tensor(x_ref)
    eps = None
    if x_dev is not None:
        eps = (torch.as_tensor(x_dev).double() - x_ref.double()).norm(dim=1)
        rel = eps / x_ref.double().norm(dim=1).clamp_min(1e-30)
        assert float(rel.max()) <= FEAT_EPS_L2_REL, f"{tag}: per-frame feature error {float(rel.max()):.3e} > {FEAT_EPS_L2_REL}"
    allowed, n_amb = unit_flip_allowed(x_ref, centers, eps)
    units, units_ref = np.asarray(units).reshape(-1), np.asarray(units_ref).reshape(-1)
    assert units.shape == units_ref.shape == (allowed.shape[0],)
    ok = allowed[np.arange(len(units)), units]
    mism = units != units_ref
    print(f"{tag}: {int(mism.sum())} mismatching frames, {n_amb} ambiguous frames of {len(units)}"
          + (f", feature error <= {float(rel.max()):.2e} (l2, relative, per frame)" if x_dev is not None else ""))
    assert ok.all(), f"{tag}: frames {np.nonzero(~ok)[0][:10]} got units {units[~ok][:10]}, reference {units_ref[~ok][:10]}"
    if max_mismatch is not None:
        assert int(mism.sum()) <= max_mismatch, f"{tag}: {int(mism.sum())} of {len(units)} units differ (allowed: {max_mismatch})"
    return int(mism.sum()), n_amb


@torch.no_grad()
def encode(sd, centers, wav, n_layers=6):
    """One utterance like the reference (B=1, data/encode.py:32): wav [1,N] -> (units [T], dense [T,768])"""
    dense = encoder(sd, conv_feature_extractor(sd, wav), n_layers)[0]
    return kmeans_assign(dense, centers), dense


def fairseq_to_hf(sd):
    """fairseq key names -> transformers.HubertModel names (its conversion script's mapping)."""
    out = {}
    for k, v in sd.items():
        if k.startswith("feature_extractor.conv_layers."):
            i, sub = k.split(".")[2], k.split(".")[3]
            rest = k.split(".")[4]
            name = "conv" if sub == "0" else "layer_norm"
            out[f"feature_extractor.conv_layers.{i}.{name}.{rest}"] = v
        elif k.startswith("post_extract_proj."):
            out["feature_projection.projection." + k.split(".")[-1]] = v
        elif k.startswith("layer_norm."):
            out["feature_projection.layer_norm." + k.split(".")[-1]] = v
        elif k == "encoder.pos_conv.0.weight_g":
            out["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = v
        elif k == "encoder.pos_conv.0.weight_v":
            out["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = v
        elif k == "encoder.pos_conv.0.bias":
            out["encoder.pos_conv_embed.conv.bias"] = v
        elif k.startswith("encoder.layer_norm."):
            out[k] = v
        elif k.startswith("encoder.layers."):
            k2 = (k.replace("self_attn_layer_norm", "layer_norm").replace("self_attn.", "attention.")
                   .replace("fc1.", "feed_forward.intermediate_dense.").replace("fc2.", "feed_forward.output_dense."))
            out[k2] = v
    return out
