"""Where the fixed cost of building the native handles goes: weight-norm fold on the host (Python),
dissc_gen_create_ex (host packing of direct / transform-domain weights + upload), the same for the HuBERT encoder and the
predictors.  python tools/create_timing.py [reps]"""
import os
import sys
import time

t0 = time.perf_counter()
import torch  # noqa: E402
t_imp = time.perf_counter() - t0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthdata as synth  # noqa: E402
import dissc_amd  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
torch.cuda.init()
torch.zeros(1, device=dev)
sd = synth.synth_generator_state_dict(seed=0)
print(f"import torch {t_imp:.3f} s")
for r in range(reps):
    g = dissc_amd.CodeGenerator(synth.VCTK_CONFIG).to(dev)
    g.load_state_dict(sd)
    g.eval()
    t = time.perf_counter()
    g.remove_weight_norm()
    t1 = time.perf_counter()
    g.prepare()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"generator rep {r}: fold {1e3 * (t1 - t):.1f} ms, create {1e3 * (t2 - t1):.1f} ms "
          f"(threads: {os.environ.get('DISSC_PACK_THREADS', 'default')})")
    del g
from dissc_amd.hubert import HubertEncoder  # noqa: E402
hsd = synth.synth_hubert_state_dict()
if hsd is not None:
    import numpy as np
    cen = torch.from_numpy(np.random.RandomState(0).standard_normal((100, 768)).astype(np.float32))
    for r in range(reps):
        t = time.perf_counter()
        e = HubertEncoder(hsd, cen, 6).to(dev)
        e._ensure()
        torch.cuda.synchronize()
        print(f"hubert rep {r}: create {1e3 * (time.perf_counter() - t):.1f} ms")
        del e
