"""CPU restatement of DISSC's predictor TRAINING step (TEST INFRASTRUCTURE).

Follows, function by function:
  len_forward_train / pitch_forward_train  <- LenPredictor.forward / PitchPredictor(.Base).forward in train() mode,
        reference model/len_predictor.py:35-52, model/pitch_predictor.py:72-94,145-166 (BatchNorm with batch
        statistics over every position of the padded batch, token-embedding masking, PositionalEncoding dropout)
  len_sum_loss                             <- LenSumLoss, reference loss/len_loss.py:16-30
  pitch_loss                               <- PitchLoss, reference loss/pitch_loss.py:6-27
  adam_step                                <- torch.optim.Adam defaults as constructed at train_len_predictor.py:35,
                                              train_f0_predictor.py:42 (betas 0.9/0.999, eps 1e-8, no weight decay)
  train_step                               <- one iteration of the loops at train_len_predictor.py:57-68 /
                                              train_f0_predictor.py:58-66
Gradients come from torch autograd on these plain-functional restatements.  The reference draws its masks from
the CUDA RNG (``torch.cuda.FloatTensor(...).uniform_()``); here -- as in the HIP engine -- the masks are explicit
inputs, which is what makes a step comparable at all.  Pinned against the imported reference modules (same masks
injected) by tests/golden/make_golden.py -> tests/golden/train.npz.
"""
import torch
import torch.nn.functional as F

BN_EPS, BN_MOM, SLOPE = 1e-5, 0.1, 0.01


def trainable_keys(sd):
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked") or k == "pe.pe")]


def _bn_train(x, sd, name, new_stats):
    """batch statistics over (batch, time); records the updated running statistics"""
    mean = x.mean(dim=(0, 2))
    var = x.var(dim=(0, 2), unbiased=False)
    n = x.shape[0] * x.shape[2]
    new_stats[name + ".running_mean"] = (1 - BN_MOM) * sd[name + ".running_mean"] + BN_MOM * mean.detach()
    new_stats[name + ".running_var"] = (1 - BN_MOM) * sd[name + ".running_var"] + BN_MOM * var.detach() * n / max(n - 1, 1)
    new_stats[name + ".num_batches_tracked"] = sd[name + ".num_batches_tracked"] + 1
    xh = (x - mean[None, :, None]) / torch.sqrt(var[None, :, None] + BN_EPS)
    return xh * sd[name + ".weight"][None, :, None] + sd[name + ".bias"][None, :, None]


def _conv(x, sd, name, pad):
    return F.conv1d(x, sd[name + ".weight"], sd[name + ".bias"], padding=pad)


def _embed(sd, seq, spk_id, keep, pe_mult=None, use_pe=False):
    """seq i64 [B,L] (pad token = last row of token_emb), spk_id i64 [B,1], keep f32 [B,L] (0 = masked token)"""
    tok = sd["token_emb.weight"]
    emb_seq = F.embedding(seq, tok, padding_idx=tok.shape[0] - 1) * keep[:, :, None]  # padding_idx = n_tokens: no grad
    emb_spk = F.embedding(spk_id, sd["spk_emb.weight"]).repeat_interleave(seq.shape[-1], dim=1)
    if use_pe:
        emb_spk = emb_spk + sd["pe.pe"][:, :seq.shape[-1]]
        if pe_mult is not None:  # dropout(p) in training: 0 or 1/(1-p) per element
            emb_spk = emb_spk * pe_mult
    return torch.cat([emb_seq, emb_spk], dim=-1).transpose(1, 2)


def len_forward_train(sd, seq, spk_id, keep, norm_mean, norm_std, new_stats):
    x = _embed(sd, seq, spk_id, keep)
    x = F.leaky_relu(_bn_train(_conv(x, sd, "cnn1", 1), sd, "bn1", new_stats), SLOPE)
    for i in range(1, 7):
        x = F.leaky_relu(_bn_train(_conv(x, sd, f"cnn1{i}", 1), sd, f"bn1{i}", new_stats), SLOPE)
    return _conv(x, sd, "cnn2", 1).squeeze(1) * norm_std + norm_mean


def pitch_forward_train(sd, seq, spk_id, keep, kind, new_stats, pe_mult=None):
    base = kind == "base"
    x = _embed(sd, seq, spk_id, keep, pe_mult, use_pe=not base)
    for n in ["cnn1"] + [f"cnn1{i}" for i in range(1, 8)]:
        x = _conv(x, sd, n, 1)
        if base:
            x = _bn_train(x, sd, "bn" + n[3:], new_stats)
        x = F.leaky_relu(x, SLOPE)
    x = _conv(x, sd, "cnn2", 1)
    if not base:
        x = _bn_train(x, sd, "bn2", new_stats)
    x = F.leaky_relu(x, SLOPE)
    c, r = _conv(x, sd, "cnn_class1", 1), _conv(x, sd, "cnn_reg1", 1)
    if base:
        c, r = _bn_train(c, sd, "bn_c1", new_stats), _bn_train(r, sd, "bn_r1", new_stats)
    cls = _conv(F.leaky_relu(c, SLOPE), sd, "cnn_class2", 0).squeeze(1)
    reg = _conv(F.leaky_relu(r, SLOPE), sd, "cnn_reg2", 0).squeeze(1)
    return cls, reg


def len_sum_loss(preds, lens, pad_idx=-1):
    diff4 = (F.avg_pool2d((preds - lens).unsqueeze(0), (1, 4)) * 4) ** 2
    mask4 = ~F.max_pool2d((lens == pad_idx).unsqueeze(0).float(), (1, 4)).bool()
    mask = lens != pad_idx
    return (mask * (preds - lens) ** 2).sum() + 0.5 * (mask4 * diff4).sum()


def pitch_loss(cls, reg, gts, spk_ids, id2mean, id2std, pad_idx=-100):
    mask = gts != pad_idx
    voiced = gts != 0
    loss1 = (mask * F.binary_cross_entropy_with_logits(cls, voiced.float(), reduction="none")).sum()
    std, mean = id2std[spk_ids.long()], id2mean[spk_ids.long()]
    loss2 = (mask * ((mean + std * reg) - (mean + std * gts)).abs() * voiced).sum()
    return 100 * loss1 + loss2


def adam_step(params, grads, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """in place; step = 1 for the first update (torch.optim.Adam, amsgrad False, weight_decay 0)"""
    for k in params:
        m[k].mul_(b1).add_(grads[k], alpha=1 - b1)
        v[k].mul_(b2).addcmul_(grads[k], grads[k], value=1 - b2)
        denom = (v[k].sqrt() / (1 - b2 ** step) ** 0.5).add_(eps)
        params[k].addcdiv_(m[k], denom, value=-lr / (1 - b1 ** step))


def train_step(kind, sd, seq, spk_id, target, keep, lr, state, norm=(0.0, 1.0), stats=None, pe_mult=None,
               pad_value=None):
    """One optimisation step.  sd: state dict (updated in place: parameters AND BatchNorm running statistics);
    state: {'m', 'v', 'step'} (created on first use).  Returns (loss, {name: grad})."""
    keys = trainable_keys(sd)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    full = dict(sd, **leaves)
    new_stats = {}
    if kind == "len":
        pred = len_forward_train(full, seq, spk_id, keep, norm[0], norm[1], new_stats)
        loss = len_sum_loss(pred, target, -1 if pad_value is None else pad_value)
    else:
        cls, reg = pitch_forward_train(full, seq, spk_id, keep, kind, new_stats, pe_mult)
        loss = pitch_loss(cls, reg, target, spk_id, stats[0], stats[1], -100 if pad_value is None else pad_value)
    grads = dict(zip(keys, torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True)))
    grads = {k: (g if g is not None else torch.zeros_like(sd[k])) for k, g in grads.items()}
    if "m" not in state:
        state.update(m={k: torch.zeros_like(sd[k]) for k in keys}, v={k: torch.zeros_like(sd[k]) for k in keys}, step=0)
    state["step"] += 1
    with torch.no_grad():
        params = {k: sd[k] for k in keys}
        adam_step(params, grads, state["m"], state["v"], state["step"], lr)
        for k, val in new_stats.items():
            sd[k] = val
    return loss.detach(), grads
