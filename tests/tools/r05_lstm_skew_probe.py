#!/usr/bin/env python3
"""Round 5: start skew of the row-block BACKWARD kernel at C4 -- by row block (round 4) against by XCD (tune key 26 bit 11) at
several skews (key 27), with the fence-free two-buffer variant (key 26 bits 8-10 = 5).  One process, interleaved rounds.
(ran at commit 9b6d159, which still had the key-26 bits 8-11 of the experiment)"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_torch_utils_network as N  # noqa: E402
from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402

dev = torch.device("cuda:0")
S, B, I, H, L = 128, 4096, 1024, 1024, 1
torch.manual_seed(0)
m = LSTM(S, B, I, H, L).to(dev)
x = torch.randn(S, B, I, device=dev, requires_grad=True)
h0, c0 = torch.randn(L, B, H, device=dev), torch.randn(L, B, H, device=dev)
N.tune_set(26, 9)
y, _ = m(x, (h0, c0))
g = torch.randn_like(y)
cfgs = [(v, xcd, sk) for v in (5,) for xcd, sk in ((0, 10), (1, 10), (1, 20), (1, 40), (0, 20), (1, 30), (1, 0))]
res = {c: [] for c in cfgs}
for rnd in range(2):
    for c in cfgs:
        v, xcd, sk = c
        N.tune_set(26, 9 | (v << 8) | (2048 if xcd else 0))
        N.tune_set(27, sk)

        def bwd():
            x.grad = None
            for p in m.parameters():
                p.grad = None
            y.backward(g, retain_graph=True)
        bwd()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            bwd()
        e1.record()
        e1.synchronize()
        res[c].append(e0.elapsed_time(e1) / 3)
N.tune_set(26, 9)
N.tune_set(27, 10)
for c in cfgs:
    print(f"variant {c[0]} skew by {'XCD' if c[1] else 'row block'} {c[2]:3d} us: backward {statistics.median(res[c]):7.2f} ms  {['%.2f' % t for t in res[c]]}")
