"""engine vs CPU oracle over a whole toy training run (same batches, same masks)"""
import sys, json, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_gpu_train import _toy_len_data
from oracle import train_ref as tr_ref
from dissc_amd.train import Trainer, init_state_dict
rs = np.random.RandomState(0)
lines = [json.loads(x) for x in _toy_len_data(96, rs)]
def dedup(u):
    v, l = [], []
    for x in u:
        if v and v[-1] == x: l[-1] += 1
        else: v.append(x); l.append(1)
    return v, l
data = [dedup(x["units"]) + (int(x["audio"][1]),) for x in lines]
allv = np.concatenate([d[1] for d in data]).astype(np.float32)
norm = (float(allv.mean()), float(allv.std(ddof=1)))
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 3e-3
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
sd0 = init_state_dict("len", 100, 4, seed=1)
tr = Trainer("len", sd0, lr, norm=norm).to("cuda:0")
sd = {k: v.clone() for k, v in sd0.items()}; st = {}
g = np.random.RandomState(3)
for ep in range(epochs):
    perm = g.permutation(96); le = lo = 0.0; n = 0
    for i in range(0, 96, 16):
        idx = perm[i:i + 16]; L = max(len(data[j][0]) for j in idx)
        seq = np.full((16, L), 100, np.int64); tgt = np.full((16, L), -1.0, np.float32); spk = np.zeros((16, 1), np.int64)
        for r, j in enumerate(idx):
            v, l, s = data[j]; seq[r, :len(v)] = v; tgt[r, :len(v)] = l; spk[r, 0] = s
        keep = (g.rand(16, L) <= 0.8).astype(np.float32)
        le += float(tr.step(seq, spk, tgt, keep=keep))
        lo += float(tr_ref.train_step("len", sd, torch.from_numpy(seq), torch.from_numpy(spk), torch.from_numpy(tgt),
                                      torch.from_numpy(keep), lr, st, norm=(torch.tensor(norm[0]), torch.tensor(norm[1])))[0])
        n += int((seq != 100).sum())
    print(ep, "engine %.4f oracle %.4f" % (le / n, lo / n))
