#!/usr/bin/env python3
"""In-process A/B of remaining launch knobs: scatter threads per block (key 2) at C5, GEMM k-depth (key 1) on the C4
LSTM forward."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi as N  # noqa: E402
dev = torch.device("cuda:0")


def t(fn, n=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


from hpc_rll.torch_utils.network.scatter_connection import ScatterConnection  # noqa: E402
B, M, C, H, W = 4096, 256, 64, 64, 64
x = torch.randn(B, M, C, device=dev)
loc = torch.stack([torch.randint(0, H, (B, M), device=dev), torch.randint(0, W, (B, M), device=dev)], -1)
for st in ("cover", "add"):
    m = ScatterConnection(B, M, C, H, W, st)
    best = {}
    for rnd in range(3):
        for tpb in (1024, 512, 256):
            N.check(N.lib.hpc_rll_tune_set(2, tpb))
            best[tpb] = min(best.get(tpb, 1e9), t(lambda: m(x, loc)))
    print(f"scatter {st}: " + "  ".join(f"tpb={k}: {v:.3f} ms" for k, v in best.items()), flush=True)
N.check(N.lib.hpc_rll_tune_set(2, 1024))
del x, loc

from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402
S, B, I, H, L = 128, 4096, 1024, 1024, 1
torch.manual_seed(0)
m = LSTM(S, B, I, H, L).to(dev)
x = torch.randn(S, B, I, device=dev)
best = {}
with torch.no_grad():
    for rnd in range(3):
        for bk in (0, 16, 32):
            N.check(N.lib.hpc_rll_tune_set(1, bk))
            best[bk] = min(best.get(bk, 1e9), t(lambda: m(x, None), n=2))
N.check(N.lib.hpc_rll_tune_set(1, 0))
print("C4 LSTM forward by GEMM k-depth: " + "  ".join(f"bk={k}: {v:.2f} ms" for k, v in best.items()), flush=True)
