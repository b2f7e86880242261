"""CPU restatement of DISSC's length / pitch predictors and infer.py's sample logic
(TEST INFRASTRUCTURE; pinned by tests/golden/pred.npz, see tests/test_oracle_golden.py).

  dedup_seq                 <- reference dataset/utils.py:14-16
  len_predictor             <- LenPredictor.forward, reference model/len_predictor.py:35-52
  len_carryover_correction  <- reference infer.py:158-172
  pitch_predictor           <- PitchPredictor.forward/infer_freq/calc_freq (+PositionalEncoding),
                               reference model/pitch_predictor.py:72-104,6-38;
                               PitchPredictorBase :145-176
  infer_sample              <- _infer_sample, reference infer.py:24-45 (pred_len + pred_pitch)
"""
import numpy as np
import torch
import torch.nn.functional as F


def dedup_seq(seq):
    """Run-length encode: (values, counts)."""
    vals, counts = [], []
    for v in seq:
        v = int(v)
        if vals and vals[-1] == v:
            counts[-1] += 1
        else:
            vals.append(v)
            counts.append(1)
    return vals, counts


def _bn_eval(x, sd, name, eps=1e-5):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], False, 0.0, eps)


def _conv(x, sd, name, pad):
    return F.conv1d(x, sd[name + ".weight"], sd[name + ".bias"], padding=pad)


@torch.no_grad()
def len_predictor(sd, seq, spk_id, norm_mean, norm_std):
    """seq int [1,L], spk_id int [1,1] -> f32 [1,L] (frames per dedup'd unit)."""
    seq = torch.as_tensor(seq).long()
    spk_id = torch.as_tensor(spk_id).long()
    emb_seq = F.embedding(seq, sd["token_emb.weight"])
    emb_spk = F.embedding(spk_id, sd["spk_emb.weight"]).repeat_interleave(seq.shape[-1], dim=1)
    x = torch.cat([emb_seq, emb_spk], dim=-1).transpose(1, 2)
    x = F.leaky_relu(_bn_eval(_conv(x, sd, "cnn1", 1), sd, "bn1"))
    for i in range(1, 7):
        x = F.leaky_relu(_bn_eval(_conv(x, sd, f"cnn1{i}", 1), sd, f"bn1{i}"))
    return _conv(x, sd, "cnn2", 1).squeeze(1) * norm_std + norm_mean


def len_carryover_correction(lens):
    """Error-diffusion rounding.  lens f32 [1,L] -> int [L].  Sequential fp32 running sum,
    torch.round (half-to-even), exactly as the reference's Python loop."""
    base = torch.round(torch.clamp(lens[0], min=1))
    a = (lens - base)[0]
    vals = []
    total = torch.zeros((), dtype=a.dtype)
    for n in a:
        total = total + n
        if total >= 1:
            vals.append(1)
            total = total - 1
        elif total <= -1:
            vals.append(-1)
            total = total + 1
        else:
            vals.append(0)
    return base.int() + torch.tensor(vals, dtype=torch.int64)


@torch.no_grad()
def pitch_predictor(sd, seq, spk_id, kind="new", norm=True, id2pitch_mean=None, id2pitch_std=None):
    """seq int [1,T], spk_id int [1,1] -> f32 [1,T]: (class_logit > 0) * regression."""
    seq = torch.as_tensor(seq).long()
    spk_id = torch.as_tensor(spk_id).long()
    T = seq.shape[-1]
    emb_seq = F.embedding(seq, sd["token_emb.weight"])
    emb_spk = F.embedding(spk_id, sd["spk_emb.weight"]).repeat_interleave(T, dim=1)
    if kind == "new":
        emb_spk = emb_spk + sd["pe.pe"][:, :T]
    x = torch.cat([emb_seq, emb_spk], dim=-1).transpose(1, 2)
    base = kind == "base"
    for n in ["cnn1"] + [f"cnn1{i}" for i in range(1, 8)]:
        x = _conv(x, sd, n, 1)
        if base:
            x = _bn_eval(x, sd, "bn" + n[3:])
        x = F.leaky_relu(x)
    x = _conv(x, sd, "cnn2", 1)
    if not base:
        x = _bn_eval(x, sd, "bn2")
    x = F.leaky_relu(x)
    c = _conv(x, sd, "cnn_class1", 1)
    r = _conv(x, sd, "cnn_reg1", 1)
    if base:
        c = _bn_eval(c, sd, "bn_c1")
        r = _bn_eval(r, sd, "bn_r1")
    cls = _conv(F.leaky_relu(c), sd, "cnn_class2", 0).squeeze(1)
    reg = _conv(F.leaky_relu(r), sd, "cnn_reg2", 0).squeeze(1)
    if not norm:
        reg = id2pitch_mean[spk_id.long()] + reg * id2pitch_std[spk_id.long()]
    return (cls > 0) * reg


@torch.no_grad()
def infer_sample(units, spk, len_sd, len_stats, pitch_sd, kind="new", norm=True, n_tokens=100,
                 id2pitch_mean=None, id2pitch_std=None):
    """units: 1-D ints.  Returns (out_units list[int], f0 list[float]) like _infer_sample's json."""
    seq = torch.as_tensor(np.asarray(units)).long()
    seq = seq[seq != n_tokens].view(1, -1)
    dd, _ = dedup_seq(seq[0].tolist())
    dd = torch.tensor(dd).unsqueeze(0)
    spk_id = torch.tensor([[int(spk)]])
    lens = len_predictor(len_sd, dd, spk_id, len_stats[0], len_stats[1])
    lens = len_carryover_correction(lens)
    out_seq = torch.repeat_interleave(dd, lens).view(1, -1)
    f0 = pitch_predictor(pitch_sd, out_seq, spk_id, kind, norm, id2pitch_mean, id2pitch_std)
    return out_seq[0].tolist(), f0[0].tolist()
