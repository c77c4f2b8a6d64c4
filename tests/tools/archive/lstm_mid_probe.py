#!/usr/bin/env python3
"""Step-path LSTM latency at moderate batch sizes (per-step = GEMM + cell kernel)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402

dev = torch.device("cuda:0")
import cabi as N  # noqa: E402
N.check(N.lib.hpc_rll_tune_set(5, int(os.environ.get('JW', '0'))))
N.check(N.lib.hpc_rll_tune_set(8, int(os.environ.get('WAVE', '1'))))
SHAPES = [(64, 3, 1792, 384, 3), (64, 16, 512, 512, 1), (64, 64, 512, 512, 1), (64, 256, 512, 512, 2), (32, 64, 256, 256, 1)]
if len(sys.argv) >= 6:
    SHAPES = [tuple(int(v) for v in sys.argv[1:6])]
for (S, B, I, H, L) in SHAPES:
    torch.manual_seed(0)
    m = LSTM(S, B, I, H, L).to(dev)
    x = torch.randn(S, B, I, device=dev, requires_grad=True)
    y, _ = m(x, None)
    g = torch.ones_like(y)
    y.backward(g, retain_graph=True)
    best = {}
    for rnd in range(3):
        for k, fn in (("fwd", lambda: m(x, None)), ("bwd", lambda: y.backward(g, retain_graph=True))):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            e1.synchronize()
            best[k] = min(best.get(k, 1e9), e0.elapsed_time(e1) / 5)
    print(f"S={S} B={B} I={I} H={H} L={L}: fwd {best['fwd']:.3f} ms ({best['fwd'] * 1e3 / S / L:.1f} us/step)  "
          f"bwd {best['bwd']:.3f} ms ({best['bwd'] * 1e3 / S / L:.1f} us/step)", flush=True)
