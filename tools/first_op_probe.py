import time, os, sys
t0=time.perf_counter(); import torch; print("import", time.perf_counter()-t0)
if len(sys.argv)>1: torch.set_num_threads(int(sys.argv[1]))
def T(name, f):
    t=time.perf_counter(); r=f(); print(f"{name}: {1e3*(time.perf_counter()-t):.1f} ms"); return r
a=T("randn small", lambda: torch.randn(8,4,3))
T("add small", lambda: a+1)
v=T("randn big", lambda: torch.randn(512,512,11)); g=torch.randn(512,1,1)
T("norm_except_dim", lambda: torch.norm_except_dim(v,2,0))
T("weight_norm big", lambda: torch._weight_norm(v,g,0))
T("weight_norm big 2", lambda: torch._weight_norm(v,g,0))
T("mul big", lambda: v*2)
print("threads", torch.get_num_threads())
