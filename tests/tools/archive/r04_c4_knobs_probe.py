#!/usr/bin/env python3
"""C4 LSTM (S = 128, B = 4096, I = H = 1024) forward / backward by the row-block kernels' skew (tune key 27) and the poll nap of the
persistent kernels (key 36), alternating in one process."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_torch_utils_network as N  # noqa: E402
from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402

dev = torch.device("cuda:0")
S, B, H = 128, 4096, 1024
m = LSTM(S, B, H, H, 1).to(dev)
x = torch.randn(S, B, H, device=dev, requires_grad=True)


def timed(fn, n=2, rounds=3):
    fn()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return statistics.median(ts)


y, _ = m(x, None)
g = torch.ones_like(y)


def bwd():
    x.grad = None
    y.backward(g, retain_graph=True)


for skew, nap in ((10, 1), (10, 8), (0, 1), (30, 1), (60, 1), (10, 1)):
    N.tune_set(27, skew)
    N.tune_set(36, nap)
    with torch.no_grad():
        tf = timed(lambda: m(x, None))
    y, _ = m(x, None)
    tb = timed(bwd)
    print(f"skew {skew:3d} nap {nap}: forward {tf:7.2f} ms  backward {tb:7.2f} ms  paths {N.lstm_last_forward_path()}/{N.lstm_last_backward_path()}", flush=True)
N.tune_set(27, 10)
N.tune_set(36, 1)
