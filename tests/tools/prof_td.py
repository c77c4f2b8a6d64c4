import sys, os, cProfile, pstats
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "di-hpc_amd"))
import torch
from hpc_rll.rl_utils.td import TDLambda
dev=torch.device("cuda:0")
T,B=1024,64
v=torch.randn(T+1,B,device=dev,requires_grad=True); r=torch.randn(T,B,device=dev); w=torch.rand(T,B,device=dev)
m=TDLambda(T,B)
def f():
    v.grad=None
    m(v,r,w).backward()
for _ in range(20): f()
torch.cuda.synchronize()
pr=cProfile.Profile(); pr.enable()
for _ in range(200): f()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
