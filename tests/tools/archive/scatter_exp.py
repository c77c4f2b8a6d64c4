import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "di-hpc_amd"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "tests"))
import torch
import cabi
import hpc_torch_utils_network as U
dev=torch.device("cuda:0")
def t(fn,n=5):
    fn(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1)/n*1e-3
B,N,H,W=4096,64,64,64
out=torch.empty(B,N,H,W,device=dev)
import itertools
for tpb, M in itertools.product((256,512,1024),(16,256)):
    assert cabi.lib.hpc_rll_tune_set(2, tpb) == 0
    x=torch.randn(B,M,N,device=dev); loc=torch.stack([torch.randint(0,H,(B,M),device=dev),torch.randint(0,W,(B,M),device=dev)],-1)
    for st in ("cover","add"):
        dt=t(lambda: U.ScatterConnectionForward([x,loc],[out],st))
        print(f"tpb={tpb} M={M:4d} {st:5s}: {dt*1e3:.3f} ms  {out.numel()*4/dt/1e9:.0f} GB/s (output bytes only)")
dt=t(lambda: out.fill_(0.0)); print(f"torch fill: {dt*1e3:.3f} ms {out.numel()*4/dt/1e9:.0f} GB/s")
