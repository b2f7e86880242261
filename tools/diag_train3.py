import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_train_oracle import initial_state
from dissc_amd.train import Trainer
rs = np.random.RandomState(5)
B, L = 32, 203
seq = np.full((B, L), 100, dtype=np.int64); tgt = np.full((B, L), -1.0, dtype=np.float32)
for b in range(B):
    n = L if b == 0 else int(rs.randint(20, L))
    seq[b, :n] = rs.randint(0, 100, size=n); tgt[b, :n] = rs.randint(1, 9, size=n)
spk = rs.randint(0, 108, size=(B, 1)).astype(np.int64)
keep = (rs.rand(B, L) <= 0.8).astype(np.float32)
outs = []
for rep in range(3):
    tr = Trainer("len", initial_state("len"), 3e-4, norm=(3.3, 2.1)).to("cuda:0")
    loss = float(tr.step(seq, spk, tgt, keep=keep))
    outs.append((loss, tr.grads()))
    torch.empty(100000000, device="cuda").fill_(float("nan"))  # dirty the allocator's memory
print([o[0] for o in outs])
for k in outs[0][1]:
    d = [(outs[0][1][k] - outs[i][1][k]).abs().max().item() for i in (1, 2)]
    if max(d) > 0: print(k, d, outs[0][1][k].abs().max().item())
