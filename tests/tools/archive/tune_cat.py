#!/usr/bin/env python3
"""Sweep the categorical row kernels' resident-blocks-per-CU knob at the C3 shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hpc_rl_utils as U
import cabi  # noqa: E402
dev = torch.device("cuda:0")
rows, N = 256 * 16384, int(os.environ.get("N", 128))
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(rows, N, device=dev, generator=g)
x2 = torch.randn(rows, N, device=dev, generator=g)
a = torch.randint(0, N, (rows,), device=dev, generator=g)
lp, ent, c1 = torch.empty(rows, device=dev), torch.empty(rows, device=dev), torch.randn(rows, device=dev)
grad = torch.empty(rows, N, device=dev)
lib, s = cabi.lib, cabi.stream_ptr(dev)
def t(fn, n=6):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / n * 1e-3
for bpc in (1, 2, 3, 4, 6, 8, 12, 16):
    assert lib.hpc_rll_tune_set(0, bpc) == 0
    # alternate two tensors so that nothing is served from the Infinity Cache
    tf = t(lambda: (lib.hpc_rll_categorical_forward(x.data_ptr(), a.data_ptr(), lp.data_ptr(), ent.data_ptr(), rows, N, s),
                    lib.hpc_rll_categorical_forward(x2.data_ptr(), a.data_ptr(), lp.data_ptr(), 0, rows, N, s))) / 2
    tb = t(lambda: lib.hpc_rll_categorical_backward(x.data_ptr(), a.data_ptr(), c1.data_ptr(), 0, c1.data_ptr(), 0, grad.data_ptr(), rows, N, s))
    by = rows * N * 4
    print(f"blocks/CU={bpc:2d}  fwd {tf*1e6:7.1f} us {by/tf/1e9:6.0f} GB/s | bwd {tb*1e6:7.1f} us {2*by/tb/1e9:6.0f} GB/s", flush=True)
