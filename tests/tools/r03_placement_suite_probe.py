#!/usr/bin/env python3
"""Does the physical placement of an OUTPUT buffer change a kernel's time (as it does for 2-wave GAE configurations)?
Each op is run into K different output allocations (previous outputs are kept alive so the caching allocator must hand out
a new block), timed per allocation with HIP events (median of 5 launches)."""
import os
import statistics
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_rl_utils as U  # noqa: E402
import hpc_torch_utils_network as NW  # noqa: E402
dev = torch.device("cuda:0")
K = 6


def timed(fn, n=5):
    fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


g = torch.Generator(device=dev).manual_seed(0)
T, B, N = 256, 16384, 128
target = torch.randn(T, B, N, device=dev, generator=g)
action = torch.randint(0, N, (T, B), device=dev, generator=g)
c1 = torch.randn(T * B, device=dev, generator=g)
one = torch.ones(1, device=dev)
outs = [torch.empty_like(target) for _ in range(K)]
import ctypes  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cabi  # noqa: E402
s = torch.cuda.current_stream().cuda_stream
ts = [timed(lambda o=o: cabi.lib.hpc_rll_categorical_backward(target.data_ptr(), action.data_ptr(), c1.data_ptr(), one.data_ptr(), None, None,
                                                               o.data_ptr(), T * B, N, s)) for o in outs]
print("categorical backward (2.15 GB in, 2.15 GB out) into 6 output allocations:", " ".join(f"{t:.0f}" for t in ts), "us", flush=True)
ins = [torch.randn(T, B, N, device=dev, generator=g) for _ in range(3)]
lp, en = torch.empty(T * B, device=dev), torch.empty(T * B, device=dev)
ts = [timed(lambda x=x: cabi.lib.hpc_rll_categorical_forward(x.data_ptr(), action.data_ptr(), lp.data_ptr(), en.data_ptr(), T * B, N, s)) for x in [target] + ins]
print("categorical forward from 4 input allocations:", " ".join(f"{t:.0f}" for t in ts), "us", flush=True)
del outs, ins, target
Bs, M, C, H, W = 4096, 256, 64, 64, 64
x = torch.randn(Bs, M, C, device=dev, generator=g)
loc = torch.stack([torch.randint(0, H, (Bs, M), device=dev, generator=g), torch.randint(0, W, (Bs, M), device=dev, generator=g)], -1)
outs = [torch.empty(Bs, C, H, W, device=dev) for _ in range(K)]
for add in (0, 1):
    ts = [timed(lambda o=o: NW.ScatterConnectionForward([x, loc], [o], "add" if add else "cover")) for o in outs]
    print(f"scatter forward ({'add' if add else 'cover'}) into 6 output allocations:", " ".join(f"{t:.0f}" for t in ts), "us", flush=True)
gx = [torch.empty_like(x) for _ in range(4)]
ts = [timed(lambda o=o: NW.ScatterConnectionBackward([outs[0], loc], [o])) for o in gx]
print("scatter backward into 4 output allocations:", " ".join(f"{t:.0f}" for t in ts), "us", flush=True)
