import sys, os, time
sys.path.insert(0, "di-hpc_amd")
import torch, numpy as np
from hpc_rll.rl_utils import padding as P
dev = torch.device("cuda:0")
n = 1 << 20
lens = torch.from_numpy(np.random.default_rng(n).integers(32, 128, n)).to(dev)
flat = torch.randn(int(lens.sum().item()), device=dev)
for _ in range(3):
    P.Padding1DPacked(flat, lens, max_len=127, group=8)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = P.Padding1DPacked(flat, lens, max_len=127, group=8)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("grouped packed pad, n=2^20, group=8: best %.3f ms median %.3f ms" % (min(ts) * 1e3, sorted(ts)[5] * 1e3))
