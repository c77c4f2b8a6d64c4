#!/usr/bin/env python3
"""Round 5 experiment: ScatterConnection cover forward at C5 with the staging lines of a LATER workgroup touched into L2 before
the stream loop (HPC_RLL_SCATTER_PF = batch offset of that workgroup; 0 = off).  One process per setting."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
from hpc_rll.torch_utils.network.scatter_connection import ScatterConnection  # noqa: E402

dev = torch.device("cuda:0")
B, M, C, H, W = 4096, 256, 64, 64, 64
torch.manual_seed(0)
x = torch.randn(B, M, C, device=dev)
loc = torch.stack([torch.randint(0, H, (B, M), device=dev), torch.randint(0, W, (B, M), device=dev)], dim=-1)
for mode in ("cover",):
    m = ScatterConnection(B, M, C, H, W, mode)
    ts = []
    y = m(x, loc)
    chk = float(y.double().sum())
    del y
    for rnd in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = m(x, loc)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
        del y
    print(f"pf={os.environ.get('HPC_RLL_SCATTER_PF', '0'):>4s} {mode}: median {statistics.median(ts):.4f} ms  {['%.4f' % t for t in sorted(ts)]} checksum {chk:.6e}")
