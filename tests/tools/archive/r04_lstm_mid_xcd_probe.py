#!/usr/bin/env python3
"""Mid-batch LSTM forward with neighbouring workgroups on one XCD (tune key 39 = 1) against launch order (0), alternating in one
process; results compared bit for bit."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_torch_utils_network as N  # noqa: E402
from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=5, rounds=7):
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


for S, B, H in ((64, 64, 1024), (64, 16, 384), (64, 64, 384), (64, 128, 512)):
    m = LSTM(S, B, H, H, 1).to(dev)
    x = torch.randn(S, B, H, device=dev)
    ref = None
    for key in (0, 1, 0, 1):
        N.tune_set(39, key)
        with torch.no_grad():
            y, _ = m(x, None)
            t = timed(lambda: m(x, None))
        if ref is None:
            ref = y.clone()
        print(f"S={S} B={B} H={H} key39={key}: forward {t * 1e3 / S:7.2f} us per step  path {N.lstm_last_forward_path()}  identical={torch.equal(ref, y)}", flush=True)
N.tune_set(39, 0)
