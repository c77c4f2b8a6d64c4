import os, sys, statistics
sys.path.insert(0, "/root/repo/di-hpc_amd")
import torch
from hpc_rll.rl_utils.td import DistNStepTD, QRDQNNStepTDError
dev = torch.device("cuda:0")
B, N, n_atom, nstep, tau = 1 << 18, 64, 51, 5, 32
g = torch.Generator(device=dev).manual_seed(0)
d = torch.softmax(torch.randn(B, N, n_atom, device=dev, generator=g), -1).requires_grad_(True)
nd = torch.softmax(torch.randn(B, N, n_atom, device=dev, generator=g), -1)
a = torch.randint(0, N, (B,), device=dev, generator=g); na = torch.randint(0, N, (B,), device=dev, generator=g)
reward = torch.randn(nstep, B, device=dev, generator=g)
done = (torch.rand(B, device=dev, generator=g) < 0.1).float()
weight = torch.rand(B, device=dev, generator=g)
m = DistNStepTD(nstep, B, N, n_atom)
def timed(fn, n, rounds=7):
    fn(); ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return ["%.1f" % t for t in sorted(ts)]
f = lambda: m(d, nd, a, na, reward, done, weight, 0.99, -10.0, 10.0)[0]
for n in (1, 5, 50):
    print("C51 grad on, burst", n, timed(f, n))
with torch.no_grad():
    for n in (1, 5, 50):
        print("C51 no_grad, burst", n, timed(f, n))
del d, nd
q = torch.randn(B, N, tau, device=dev, generator=g, requires_grad=True); nq = torch.randn(B, N, tau, device=dev, generator=g)
m2 = QRDQNNStepTDError(tau, nstep, B, N)
f2 = lambda: m2(q, nq, a, na, reward, done, 0.99, weight)[0]
for n in (1, 5, 50):
    print("QR grad on, burst", n, timed(f2, n))
