#!/usr/bin/env python3
"""Two persistent/wavefront LSTM passes on two streams at once: do they stay co-resident (no starvation trap)?"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402
dev = torch.device("cuda:0")
for (S, B, I, H, L) in [(64, 3, 1792, 384, 3), (64, 4, 512, 512, 2), (32, 8, 64, 1024, 1)]:
    torch.manual_seed(0)
    ms = [LSTM(S, B, I, H, L).to(dev) for _ in range(2)]
    xs = [torch.randn(S, B, I, device=dev, requires_grad=True) for _ in range(2)]
    ref = []
    for m, x in zip(ms, xs):
        y, _ = m(x, None)
        y.sum().backward()
        ref.append((y.detach().clone(), x.grad.clone()))
        x.grad = None
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    for it in range(20):
        outs = []
        for m, x, st in zip(ms, xs, streams):
            with torch.cuda.stream(st):
                y, _ = m(x, None)
                y.sum().backward()
                outs.append((y, x.grad))
        torch.cuda.synchronize()
        for (y, g), (ry, rg) in zip(outs, ref):
            assert torch.equal(y, ry) and torch.equal(g, rg)
        for x in xs:
            x.grad = None
    print(f"S={S} B={B} H={H} L={L}: 20 concurrent iterations on 2 streams OK (bit-identical to serial)", flush=True)
