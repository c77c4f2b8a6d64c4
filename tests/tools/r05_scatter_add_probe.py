#!/usr/bin/env python3
"""Round 5: ScatterConnection `add` forward at C5 -- the cells-per-thread kernel behind the index launch (tune key 18 = 0: the
rule's choice before the staging prefetch) against the LDS-staged kernel with in-kernel chain tables at 32 / 64 channels per
workgroup (key 18 = 32 / 64), whose stream loop is unrolled and whose staging lines are prefetched into L2 by an earlier
workgroup.  Interleaved rounds, HIP events; bit-identical outputs expected."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_torch_utils_network as N  # noqa: E402
from hpc_rll.torch_utils.network.scatter_connection import ScatterConnection  # noqa: E402

dev = torch.device("cuda:0")
B, M, C, H, W = 4096, 256, 64, 64, 64
torch.manual_seed(0)
x = torch.randn(B, M, C, device=dev)
loc = torch.stack([torch.randint(0, H, (B, M), device=dev), torch.randint(0, W, (B, M), device=dev)], dim=-1)
for mode in ("add", "cover"):
    m = ScatterConnection(B, M, C, H, W, mode)
    res, outs = {}, {}
    keys = (0, 32, 64) if mode == "add" else (0, 32)
    for rnd in range(3):
        for k in keys:
            N.tune_set(18, k)
            y = m(x, loc)
            torch.cuda.synchronize()
            outs.setdefault(k, y.clone())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                y = m(x, loc)
            e1.record()
            e1.synchronize()
            res.setdefault(k, []).append(e0.elapsed_time(e1) / 10)
            del y
    N.tune_set(18, 0)
    N.tune_set(17, 0)                                   # the cells-per-thread kernel behind the index launch
    ref = m(x, loc)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        y = m(x, loc)
    t1.record()
    t1.synchronize()
    N.tune_set(17, 1)
    print(f"{mode} cells-per-thread kernel (key 17 = 0): {t0.elapsed_time(t1) / 10:.4f} ms; LDS kernel output identical to it: {torch.equal(ref, outs[keys[0]])}")
    del ref, y
    for k in keys:
        print(f"{mode} key18={k}: {statistics.median(res[k]):.4f} ms {['%.4f' % t for t in res[k]]} identical to key 0: {torch.equal(outs[k], outs[keys[0]])}")
