#!/usr/bin/env python3
"""C4 LSTM backward under GEMM knobs: tune key 1 (BK) x key 6 (split-K target of the weight-gradient GEMMs)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi as N  # noqa: E402
from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402
dev = torch.device("cuda:0")
S, B, I, H, L = 128, 4096, 1024, 1024, 1
torch.manual_seed(0)
m = LSTM(S, B, I, H, L).to(dev)
x = torch.randn(S, B, I, device=dev, requires_grad=True)
h0 = torch.randn(L, B, H, device=dev)
c0 = torch.randn(L, B, H, device=dev)
for bk, target, t128 in [(0, 768, 0), (0, 768, 1), (16, 768, 1), (32, 768, 1), (0, 1024, 1), (16, 1024, 1), (0, 768, 0)]:
    N.check(N.lib.hpc_rll_tune_set(1, bk))
    N.check(N.lib.hpc_rll_tune_set(6, target))
    N.check(N.lib.hpc_rll_tune_set(7, t128))
    y, _ = m(x, (h0, c0))
    g = torch.ones_like(y)
    y.backward(g, retain_graph=True)
    ts = []
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y.backward(g, retain_graph=True)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"bk={bk} target={target} tile128={t128}: bwd {min(ts):.1f} ms", flush=True)
    del y, g
