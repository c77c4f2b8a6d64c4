import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, time
import bench_suite as S
dev = torch.device("cuda:0")
from hpc_rll.rl_utils.vtrace import VTrace
from hpc_rll.rl_utils.gae import GAE
def preroll(sec):
    v = torch.randn(1025, 65536, device=dev, requires_grad=True); r = torch.randn(1024, 65536, device=dev, requires_grad=True)
    gg = torch.randn(1024, 65536, device=dev); m = GAE(1024, 65536)
    t0 = time.time()
    while time.time() - t0 < sec:
        for _ in range(200):
            v.grad = r.grad = None
            m(v, r).backward(gg)
        torch.cuda.synchronize()
preroll(4)
T,B,n=256,16384,128
g = torch.Generator(device=dev).manual_seed(0)
xt = torch.randn(T, B, n, device=dev, generator=g); xb = torch.randn(T, B, n, device=dev, generator=g)
a = torch.randint(0, n, (T, B), device=dev, generator=g)
value = torch.randn(T + 1, B, device=dev, generator=g); reward = torch.randn(T, B, device=dev, generator=g)
vt = VTrace(T,B,n)
def t(fn, k):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
print("no-grad inputs, k=10:", [round(t(lambda: vt(xt, xb, a, value, reward),10)) for _ in range(5)])
print("no-grad inputs, k=5 :", [round(t(lambda: vt(xt, xb, a, value, reward),5)) for _ in range(5)])
xt.requires_grad_(True); value.requires_grad_(True)
print("grad inputs, k=5    :", [round(t(lambda: vt(xt, xb, a, value, reward),5)) for _ in range(5)])
print("grad inputs+sum, k=5:", [round(t(lambda: sum(vt(xt, xb, a, value, reward)),5)) for _ in range(5)])
del xt, xb
torch.cuda.empty_cache()
S.suite_c3()
