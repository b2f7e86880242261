import sys, ctypes, numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_train_oracle import initial_state
from dissc_amd.train import Trainer
from dissc_amd._lib import lib, check
from oracle import train_ref as tr
lib.dissc_train_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
g = np.load("tests/golden/train.npz")
kind = "len"
pre = f"{kind}/s0/"
seq, tgt, spk, keep = (torch.from_numpy(g[pre + n]) for n in ("seq", "tgt", "spk", "keep"))
T = Trainer(kind, initial_state(kind), 3e-4, norm=(3.3, 2.1)).to("cuda:0")
T.step(seq, spk, tgt, keep=keep)
B, L = seq.shape; ld = (L + 3) // 4 * 4
def rd(layer, which, C):
    out = torch.empty(B * C * ld)
    check(lib.dissc_train_debug_read(T._h, layer, which, out.data_ptr(), out.numel(), None), "dbg")
    return out.view(B, C, ld)[:, :, :L]
# oracle intermediates with autograd
sd = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in initial_state(kind).items()}
ns = {}
x = tr._embed(sd, seq, spk, keep); x.retain_grad(); acts = [x]; zs = []
names = ["cnn1"] + [f"cnn1{i}" for i in range(1, 7)]
for i, n in enumerate(names):
    z = tr._conv(acts[-1], sd, n, 1); z.retain_grad(); zs.append(z)
    a = F.leaky_relu(tr._bn_train(z, sd, "bn" + n[3:], ns), 0.01); a.retain_grad(); acts.append(a)
out = tr._conv(acts[-1], sd, "cnn2", 1).squeeze(1)
loss = tr.len_sum_loss(out * 2.1 + 3.3, tgt); loss.backward()
for i in range(7):
    for which, name, ref in ((0, "z", zs[i]), (1, "a", acts[i + 1]), (2, "da", acts[i + 1].grad), (3, "dz", zs[i].grad)):
        got = rd(i, which, 128)
        print(f"layer {i} {name}: rel {float((got - ref.detach()).norm() / ref.detach().norm()):.3e}")
print("dx0 rel", float((rd(-1, 2, 64) - acts[0].grad).norm() / acts[0].grad.norm()))
gr = T.grads()
hb, ob = gr["bn16.bias"].numpy(), sd["bn16.bias"].grad.numpy()
print("bn16.bias hip", hb[:6], "\n          ref", ob[:6], "\n ratio", (hb / ob)[:10])
da6 = rd(6, 2, 128); a6 = rd(6, 1, 128)
dy = da6 * torch.where(a6 > 0, torch.ones_like(a6), torch.full_like(a6, 0.01))
print("s1 from dumped buffers", dy.sum((0, 2))[:6].numpy())
z6 = rd(6, 0, 128); dz6 = rd(6, 3, 128)
ref = zs[6].grad
err = (dz6 - ref)
print("dz6 err per channel (first 8):", err.abs().amax((0, 2))[:8].numpy(), "ref scale", ref.abs().amax((0,2))[:8].numpy())
print("err const over (b,t)? std/mean of err per channel:", (err.std((0,2)) / (err.mean((0,2)).abs()+1e-12))[:8].numpy())
print("err vs xh corr: ", [float(torch.corrcoef(torch.stack([err[:,c].reshape(-1), z6[:,c].reshape(-1)]))[0,1]) for c in range(4)])
hg, og = gr["bn16.weight"].numpy(), sd["bn16.weight"].grad.numpy()
print("dgamma ratio", (hg/og)[:8])
pc = err.abs().amax((0, 2)) / ref.abs().amax((0, 2))
bad = torch.nonzero(pc > 1e-4).flatten().tolist()
print("bad channels:", bad, [round(float(pc[c]), 4) for c in bad][:20])
c = bad[0]
print("chan", c, "err[b=0]:", err[0, c, :12].numpy(), "\nref:", ref[0, c, :12].numpy(), "\nerr[b=1]", err[1, c, :12].numpy())
print("gamma", float(sd["bn16.weight"][c]), "dgamma ratio", float(hg[c] / og[c]), "dbeta ratio", float(hb[c] / ob[c]))
