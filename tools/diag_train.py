import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_train_oracle import initial_state, compact
from dissc_amd.train import Trainer
g = np.load("tests/golden/train.npz")
for kind in ("len", "new", "base"):
    lr = float(g[f"{kind}/lr"])
    stats = (torch.from_numpy(g["id2pitch_mean"]), torch.from_numpy(g["id2pitch_std"]))
    tr = Trainer(kind, initial_state(kind), lr, norm=(3.3, 2.1), stats=stats).to("cuda:0")
    pre = f"{kind}/s0/"
    seq, tgt, spk, keep = (torch.from_numpy(g[pre + n]) for n in ("seq", "tgt", "spk", "keep"))
    pe_mult = torch.from_numpy(g[pre + "pe_mult"]) if kind == "new" else None
    loss = float(tr.step(seq, spk, tgt, keep=keep, pe_mult=pe_mult))
    print(kind, "loss", loss, float(g[pre + "loss"]))
    for k, gv in tr.grads().items():
        a, b = compact(gv.numpy()).astype(np.float64), np.asarray(g[pre + "grad/" + k], dtype=np.float64)
        print(f"  {k:24s} rel l2 {np.linalg.norm(a-b)/(np.linalg.norm(b)+1e-30):.3e}  max|want| {np.abs(b).max():.3e}")
