#!/usr/bin/env python3
"""Is the categorical forward (V-trace / UPGO / PPO heads; 2.15 GB streamed once) sensitive to WHERE its input lives?
Eight (T,B,N) = (256,16384,128) tensors allocated in a row, each timed on its own (median of 7 launches); then everything
is released to the driver (empty_cache) and the exercise repeated."""
import os
import statistics
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi  # noqa: E402
dev = torch.device("cuda:0")
T, B, N = 256, 16384, 128
s = torch.cuda.current_stream().cuda_stream


def timed(fn, n=7):
    fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


g = torch.Generator(device=dev).manual_seed(0)
action = torch.randint(0, N, (T, B), device=dev, generator=g)
lp, en = torch.empty(T * B, device=dev), torch.empty(T * B, device=dev)
pre = torch.empty(1 << 28, device=dev)
for _ in range(300):
    pre.add_(1.0)
del pre
for rnd in range(3):
    xs = [torch.randn(T, B, N, device=dev, generator=g) for _ in range(8)]
    ts = [timed(lambda x=x: cabi.lib.hpc_rll_categorical_forward(x.data_ptr(), action.data_ptr(), lp.data_ptr(), en.data_ptr(), T * B, N, s)) for x in xs]
    outs = [torch.empty_like(xs[0]) for _ in range(3)]
    c1 = torch.randn(T * B, device=dev, generator=g)
    one = torch.ones(1, device=dev)
    tb = [timed(lambda x=x, o=o: cabi.lib.hpc_rll_categorical_backward(x.data_ptr(), action.data_ptr(), c1.data_ptr(), one.data_ptr(), None, None,
                                                                      o.data_ptr(), T * B, N, s)) for x, o in ((xs[0], outs[0]), (xs[0], outs[1]), (xs[1], outs[0]), (xs[5], outs[2]))]
    rd = [timed(lambda x=x: torch.sum(x)) for x in xs[:4]]
    print(f"round {rnd}: forward per input {' '.join(f'{t:.0f}' for t in ts)} us | backward (in0,out0) (in0,out1) (in1,out0) (in5,out2): "
          f"{' '.join(f'{t:.0f}' for t in tb)} | addresses {' '.join(hex(x.data_ptr() >> 21) for x in xs[:4])}", flush=True)
    del xs, outs
    if rnd == 0:
        torch.cuda.empty_cache()
