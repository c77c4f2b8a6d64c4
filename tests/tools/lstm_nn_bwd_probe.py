#!/usr/bin/env python3
"""LSTM backward: products against transposed weight copies (NN, tune key 11 = 1) vs the NT form, interleaved."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi as N  # noqa: E402
from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402
dev = torch.device("cuda:0")
SH = [(128, 4096, 1024, 1024, 1), (64, 16, 512, 512, 1), (64, 64, 512, 512, 1), (64, 256, 512, 512, 2), (32, 64, 256, 256, 1), (64, 1024, 512, 512, 1)]
if os.environ.get('ONLY_C4'):
    SH = SH[:1] + [(64, 8192, 512, 512, 1), (32, 2048, 2048, 2048, 1)]
for (S, B, I, H, L) in SH:
    torch.manual_seed(0)
    m = LSTM(S, B, I, H, L).to(dev)
    x = torch.randn(S, B, I, device=dev, requires_grad=True)
    best = {}
    for rnd in range(3):
        for flag in (1, 0):
            N.check(N.lib.hpc_rll_tune_set(int(os.environ.get('KEY', '11')), flag))
            y, _ = m(x, None)
            g = torch.ones_like(y)
            y.backward(g, retain_graph=True)
            n = 2 if B >= 4096 else 5
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                y.backward(g, retain_graph=True)
            e1.record()
            e1.synchronize()
            best[flag] = min(best.get(flag, 1e9), e0.elapsed_time(e1) / n)
            del y, g
    print(f"S={S} B={B} I={I} H={H} L={L}: bwd NN {best[1]:.3f} ms   NT {best[0]:.3f} ms", flush=True)
N.check(N.lib.hpc_rll_tune_set(int(os.environ.get('KEY', '11')), 1))
