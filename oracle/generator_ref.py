"""CPU restatement of the DISSC speech-resynthesis generator (TEST INFRASTRUCTURE).

Plain PyTorch CPU fp32 ops, one utterance at a time like the reference
(B=1, reference sr/inference.py:178).  Pinned against the imported reference
in this container by tests/golden/make_golden.py -> tests/golden/gen_*.npz
(see tests/test_oracle_golden.py).

Follows, function by function:
  fold_weight_norm      <- torch.nn.utils.remove_weight_norm as called at
                           reference sr/models.py:43-47,116-122
  embed_concat          <- CodeGenerator.forward, reference sr/models.py:179-215
  generator_forward     <- Generator.forward, reference sr/models.py:98-114
  resblock1             <- ResBlock1.forward, reference sr/models.py:34-41
  wav_postprocess       <- generate()/inference(), reference
                           sr/inference.py:73-75,205-206
"""
import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # reference sr/models.py:13


def fold_weight_norm(weight_g, weight_v):
    """w = g * v / ||v||, norm over every dim but 0 (weight_norm default dim=0).

    For ConvTranspose1d the weight is [Cin, Cout, k], so g is per *input*
    channel (reference sr/models.py:83-86 wraps ConvTranspose1d in weight_norm).
    """
    # torch._weight_norm is the primitive remove_weight_norm itself evaluates;
    # using it keeps the folded weights bit-identical to the reference's
    # (an explicit v*(g/||v||) differs in the last ulp).
    return torch._weight_norm(weight_v, weight_g, 0)


def fold_state_dict(sd):
    """{'x.weight_g','x.weight_v','x.bias'} -> {'x.weight','x.bias'}; others kept."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            base = k[: -len(".weight_g")]
            out[base + ".weight"] = fold_weight_norm(v.float(), sd[base + ".weight_v"].float())
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = v.float()
    return out


def get_padding(kernel_size, dilation=1):
    # reference sr/utils.py:44-45
    return int((kernel_size * dilation - dilation) / 2)


def embed_concat(w, code, f0, spkr):
    """code i64 [1,T], f0 f32 [1,1,T], spkr i64 [1,1] -> x [1,257,T].

    Channel order [code-emb 128 | f0 1 | spkr-emb 128]
    (reference sr/models.py:189,207-215)."""
    x = F.embedding(code, w["dict.weight"]).transpose(1, 2)

    def upsample(sig, n):
        # reference _upsample (sr/models.py:158-177): integer repeat along time only
        if n % sig.shape[-1] != 0:
            raise NotImplementedError("Padding condition signal - misalignment between condition features.")
        return sig.repeat_interleave(n // sig.shape[-1], dim=-1)

    # reference sr/models.py:206-210: the SHORTER of the two streams is repeated up to the longer one -- the embedded code stream when
    # it is shorter than f0 (a coarser unit rate), else f0
    if x.shape[-1] < f0.shape[-1]:
        x = upsample(x, f0.shape[-1])
    elif f0.shape[-1] != x.shape[-1]:
        f0 = upsample(f0, x.shape[-1])
    T = x.shape[-1]
    x = torch.cat([x, f0], dim=1)
    s = F.embedding(spkr, w["spkr.weight"]).transpose(1, 2)  # [1,128,1]
    x = torch.cat([x, s.expand(-1, -1, T)], dim=1)
    return x


def resblock1(w, prefix, x, k, dilations=(1, 3, 5)):
    for m, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, w[f"{prefix}.convs1.{m}.weight"], w[f"{prefix}.convs1.{m}.bias"],
                      padding=get_padding(k, d), dilation=d)
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, w[f"{prefix}.convs2.{m}.weight"], w[f"{prefix}.convs2.{m}.bias"],
                      padding=get_padding(k, 1))
        x = xt + x
    return x


def generator_forward(w, h, x, taps=None):
    """x [1,in_dim,T] -> wav [1,1,T*prod(upsample_rates)].  ``taps`` (dict)
    receives the per-stage activations when given."""
    nk = len(h["resblock_kernel_sizes"])
    x = F.conv1d(x, w["conv_pre.weight"], w["conv_pre.bias"], padding=3)
    if taps is not None:
        taps["conv_pre"] = x
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, w[f"ups.{i}.weight"], w[f"ups.{i}.bias"],
                               stride=u, padding=(k - u) // 2)
        if taps is not None:
            taps[f"up{i}"] = x
        xs = None
        for j, (rk, rd) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            r = resblock1(w, f"resblocks.{i * nk + j}", x, rk, tuple(rd))
            xs = r if xs is None else xs + r
        x = xs / nk
        if taps is not None:
            taps[f"mrf{i}"] = x
    x = F.leaky_relu(x)  # default slope 0.01 (reference sr/models.py:110)
    x = F.conv1d(x, w["conv_post.weight"], w["conv_post.bias"], padding=3)
    return torch.tanh(x)


@torch.no_grad()
def code_generator(w, h, code, f0, spkr, lengths=None, taps=None):
    """Batched front end with the reference's semantics: every utterance is run
    alone at its true length (the reference never batches), outputs are right
    padded with zeros.  code [B,T] i64, f0 [B,1,T], spkr [B,1] -> [B,1,hop*T]."""
    code = torch.as_tensor(code)
    f0 = torch.as_tensor(f0)
    spkr = torch.as_tensor(spkr)
    B, T = code.shape
    hop = int(np.prod(h["upsample_rates"]))
    out = torch.zeros(B, 1, hop * T)
    for b in range(B):
        n = T if lengths is None else int(lengths[b])
        if n == 0:
            continue
        x = embed_concat(w, code[b:b + 1, :n], f0[b:b + 1, :, :n], spkr[b:b + 1])
        y = generator_forward(w, h, x, taps if (taps is not None and b == 0) else None)
        out[b, :, : hop * n] = y[0]
    return out


MAX_WAV_VALUE = 32768.0  # reference sr/dataset.py:24


def wav_postprocess(y):
    """float waveform in (-1,1) -> the float32 samples the reference writes.

    reference sr/inference.py:73-75: (y*32768).astype('int16') -- C-style
    truncation toward zero; tanh keeps |y|<=1 so only +1.0 can hit 32768,
    which numpy wraps to -32768.  :206: librosa.util.normalize(float32) =
    x / max|x| (left untouched when the peak is below float32 tiny)."""
    a = (np.asarray(y, dtype=np.float32) * np.float32(MAX_WAV_VALUE))
    a = np.trunc(a).astype(np.int64)
    a = ((a + 32768) % 65536 - 32768).astype(np.int16)
    x = a.astype(np.float32)
    peak = np.max(np.abs(x)) if x.size else np.float32(0)
    if peak >= np.finfo(np.float32).tiny:
        x = x / peak
    return x
