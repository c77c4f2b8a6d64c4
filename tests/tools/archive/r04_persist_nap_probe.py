#!/usr/bin/env python3
"""The nap between polls of the small-batch persistent LSTM kernels (tune key 36: 8 = shipped since round 2, 1 = 64 cycles) at the
reference's test shape (S=64, B=3, I=1792, H=384, L=3) and at B=4, H=1024."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_torch_utils_network as N  # noqa: E402
from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=5, rounds=5):
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


for S, B, I, H, L in ((64, 3, 1792, 384, 3), (64, 4, 256, 1024, 1), (64, 2, 64, 512, 2)):
    m = LSTM(S, B, I, H, L).to(dev)
    x = torch.randn(S, B, I, device=dev, requires_grad=True)
    y, _ = m(x, None)
    g = torch.ones_like(y)

    def bwd():
        x.grad = None
        y.backward(g, retain_graph=True)
    for nap in (8, 1, 8, 1):
        N.tune_set(36, nap)
        print(f"S={S} B={B} I={I} H={H} L={L} nap {nap}: fwd {timed(lambda: m(x, None)):.4f} ms  bwd {timed(bwd):.4f} ms  paths {N.lstm_last_forward_path()}/{N.lstm_last_backward_path()}", flush=True)
    N.tune_set(36, 1)
    for rep in (4, 8, 16, 2, 4):
        N.tune_set(4, rep)
        print(f"S={S} B={B} I={I} H={H} L={L} nap 1 rep {rep}: fwd {timed(lambda: m(x, None)):.4f} ms  bwd {timed(bwd):.4f} ms", flush=True)
    N.tune_set(4, 4)
