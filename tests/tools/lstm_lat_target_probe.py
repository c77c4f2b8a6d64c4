#!/usr/bin/env python3
"""Mid-batch LSTM: workgroups the latency-regime split-K aims for (tune key 13), interleaved rounds."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi as N  # noqa: E402
from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402
dev = torch.device("cuda:0")
KEY = int(os.environ.get('KEY', '13'))
VALS = [int(v) for v in os.environ.get('VALS', '64,128,256,512').split(',')]
SHAPES = [(64, 16, 512, 512, 1), (64, 64, 512, 512, 1), (64, 256, 512, 512, 1), (64, 64, 256, 256, 1), (64, 128, 1024, 1024, 1), (64, 1024, 512, 512, 1)]
if os.environ.get('BIG'):
    SHAPES = [(64, 1024, 512, 512, 1), (64, 2048, 512, 512, 1), (64, 512, 1024, 1024, 1), (64, 2048, 1024, 1024, 1), (128, 4096, 1024, 1024, 1)]
for (S, B, I, H, L) in SHAPES:
    torch.manual_seed(0)
    m = LSTM(S, B, I, H, L).to(dev)
    x = torch.randn(S, B, I, device=dev, requires_grad=True)
    best = {}
    for rnd in range(3):
        for tgt in VALS:
            N.check(N.lib.hpc_rll_tune_set(KEY, tgt))
            y, _ = m(x, None)
            g = torch.ones_like(y)
            y.backward(g, retain_graph=True)
            for k, fn in (("fwd", lambda: m(x, None)), ("bwd", lambda: y.backward(g, retain_graph=True))):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(2 if B >= 4096 else 5):
                    fn()
                e1.record()
                e1.synchronize()
                best[(tgt, k)] = min(best.get((tgt, k), 1e9), e0.elapsed_time(e1) / (2 if B >= 4096 else 5))
            del y, g
    print(f"S={S} B={B} H={H}: " + "  ".join(f"{t}: {best[(t, 'fwd')]:.3f}/{best[(t, 'bwd')]:.3f}" for t in VALS), flush=True)
N.check(N.lib.hpc_rll_tune_set(KEY, {13: 256, 14: 8}.get(KEY, 0)))
