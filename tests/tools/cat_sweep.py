#!/usr/bin/env python3
"""Categorical row kernels across action-space sizes N (rows = 2^22): achieved bandwidth vs algorithmic bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hpc_rl_utils as U
import cabi  # noqa: E402
dev = torch.device("cuda:0")
lib, s = cabi.lib, cabi.stream_ptr(dev)
def t(fn, n=5):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / n * 1e-3
for N in (2, 4, 6, 9, 18, 32, 64, 100, 128, 256, 512, 1000, 2048, 5000):
    rows = max(1 << 14, min(1 << 22, (1 << 29) // N))
    x = torch.randn(rows, N, device=dev); x2 = torch.randn(rows, N, device=dev)
    a = torch.randint(0, N, (rows,), device=dev)
    lp, ent, c1 = torch.empty(rows, device=dev), torch.empty(rows, device=dev), torch.randn(rows, device=dev)
    grad = torch.empty(rows, N, device=dev)
    tf = t(lambda: (lib.hpc_rll_categorical_forward(x.data_ptr(), a.data_ptr(), lp.data_ptr(), ent.data_ptr(), rows, N, s),
                    lib.hpc_rll_categorical_forward(x2.data_ptr(), a.data_ptr(), lp.data_ptr(), ent.data_ptr(), rows, N, s))) / 2
    tb = t(lambda: lib.hpc_rll_categorical_backward(x.data_ptr(), a.data_ptr(), c1.data_ptr(), 0, c1.data_ptr(), 0, grad.data_ptr(), rows, N, s))
    bf, bb = rows * (4 * N + 16), rows * (8 * N + 16)
    print(f"N={N:5d} rows={rows:8d} fwd {tf*1e6:8.1f} us {bf/tf/1e9:6.0f} GB/s | bwd {tb*1e6:8.1f} us {bb/tb/1e9:6.0f} GB/s", flush=True)
    del x, x2, a, lp, ent, c1, grad
