import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synthdata as synth
from dissc_amd import predictors as P
from dissc_amd.hubert import HubertEncoder
dev='cuda:0'
enc = HubertEncoder(synth.synth_hubert_state_dict(6), synth.synth_kmeans_centers(), 6).to(dev)
lm = P.LenPredictor(100, 108).to(dev); lm.load_state_dict(synth.synth_len_state_dict(100, 108))
n=160000; utts=32
waves=[torch.from_numpy(synth.synth_waveform(n, seed=i)).to(dev) for i in range(utts)]
e = enc(torch.stack(waves), n_samples=torch.full((utts,), n, dtype=torch.int32), want_dense=False)
units, frames = e["units"], e["frames"].to(dev)
spk=torch.full((utts,1),6,dtype=torch.int64)
vals,_,nn_=P.dedup(units,frames)
print("frames",frames.tolist()[:4],"dedup n",nn_.tolist())
lm.norm_mean, lm.norm_std = torch.tensor(0.0), torch.tensor(1.0)
raw=lm(vals,spk,lengths=nn_)
valid=torch.arange(raw.shape[1],device=dev)[None,:]<nn_[:,None]
r=raw[valid].double()
print("raw mean/std",float(r.mean()),float(r.std()),"min/max",float(r.min()),float(r.max()))
pu=[(float(raw[b,:nn_[b]].mean()),float(raw[b,:nn_[b]].std())) for b in range(utts)]
print("per-utt mean/std",[(round(a,2),round(b,2)) for a,b in pu[:8]])
