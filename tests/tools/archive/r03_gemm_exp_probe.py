#!/usr/bin/env python3
"""Round 3: two cheap experiments on the 256x256 fp32 MFMA tile (tune key 23), in-process A/B at the C4 LSTM and the
standalone products: bit 0 = s_setprio(1) around the MFMA clusters, bit 1 = k-depth 32."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_torch_utils_network as U  # noqa: E402
from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402
dev = torch.device("cuda:0")


def t(fn, n=3):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


a = torch.randn(4096, 4096, device=dev)
b = torch.randn(4096, 4096, device=dev)
c = torch.empty(4096, 4096, device=dev)
ref = None
best = {}
for rnd in range(3):
    for k in (0, 1, 2, 3):
        U.tune_set(23, k)
        best[k] = min(best.get(k, 1e9), t(lambda: U.gemm_f32(a, b, out=c), n=10))
        if ref is None:
            ref = c.clone()
        assert torch.equal(ref, c), k
print("gemm 4096^3 by key 23: " + "  ".join(f"{k}: {v*1e3:.1f} us ({2*4096**3/v/1e9:.1f} TF/s)" for k, v in best.items()), flush=True)
a2 = torch.randn(4096, 1024, device=dev)
b2 = torch.randn(1024, 4096, device=dev)
best = {}
for rnd in range(3):
    for k in (0, 1, 2, 3):
        U.tune_set(23, k)
        best[k] = min(best.get(k, 1e9), t(lambda: U.gemm_f32(a2, b2, out=c), n=20))
print("gemm 4096x4096x1024 by key 23: " + "  ".join(f"{k}: {v*1e3:.1f} us ({2*4096*4096*1024/v/1e9:.1f} TF/s)" for k, v in best.items()), flush=True)
del a, b, c, a2, b2
S, B, I, H, L = 128, 4096, 1024, 1024, 1
torch.manual_seed(0)
m = LSTM(S, B, I, H, L).to(dev)
x = torch.randn(S, B, I, device=dev, requires_grad=True)
bf, bb = {}, {}
for rnd in range(3):
    for k in (0, 1, 2, 3):
        U.tune_set(23, k)
        bf[k] = min(bf.get(k, 1e9), t(lambda: m(x, None), n=2))
        y, _ = m(x, None)
        g = torch.ones_like(y)

        def bwd():
            x.grad = None
            for p in m.parameters():
                p.grad = None
            y.backward(g, retain_graph=True)
        bb[k] = min(bb.get(k, 1e9), t(bwd, n=2))
        del y
U.tune_set(23, 0)
print("C4 LSTM forward by key 23: " + "  ".join(f"{k}: {v:.2f} ms" for k, v in bf.items()), flush=True)
print("C4 LSTM backward by key 23: " + "  ".join(f"{k}: {v:.2f} ms" for k, v in bb.items()), flush=True)
