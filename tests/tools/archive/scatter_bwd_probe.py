#!/usr/bin/env python3
"""ScatterConnection backward at the C5 shape: LDS budget per workgroup (tune key 9)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi as N  # noqa: E402
from hpc_rll.torch_utils.network.scatter_connection import ScatterConnection  # noqa: E402
dev = torch.device("cuda:0")
B, M, C, H, W = 4096, 256, 64, 64, 64
x = torch.randn(B, M, C, device=dev, requires_grad=True)
loc = torch.stack([torch.randint(0, H, (B, M), device=dev), torch.randint(0, W, (B, M), device=dev)], -1)
m = ScatterConnection(B, M, C, H, W, "cover")
out = m(x, loc)
go = torch.randn_like(out)
for rnd in range(2):
    for kb in (64, 32, 96, 128):
        N.check(N.lib.hpc_rll_tune_set(9, kb))
        def bwd():
            x.grad = None
            out.backward(go, retain_graph=True)
        bwd()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            bwd()
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1) / 5
        print(f"lds={kb} KB: bwd {t:.3f} ms  ({(4*B*C*H*W + 4*B*M*C)/t/1e6:.0f} GB/s)", flush=True)
