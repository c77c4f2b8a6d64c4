#!/usr/bin/env python3
"""Latency probe of the small-batch LSTM (reference test shape): forward/backward time with the persistent path on
and off; with HPC_RLL_LSTM_PROFILE=1 the library prints the per-phase time of workgroup 0."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi as N  # noqa: E402
from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(64, 3, 1792, 384, 3)] if len(sys.argv) < 6 else [tuple(int(v) for v in sys.argv[1:6])]
for (S, B, I, H, L) in shapes:
    torch.manual_seed(0)
    m = LSTM(S, B, I, H, L).to(dev)
    x = torch.randn(S, B, I, device=dev, requires_grad=True)
    stride = int(os.environ.get("XS", "4"))
    N.check(N.lib.hpc_rll_tune_set(4, stride))
    N.check(N.lib.hpc_rll_tune_set(5, int(os.environ.get("JW", "0"))))
    best = {}
    for rnd in range(4):                      # modes interleaved: no clock-ramp / ordering bias
        for mode in (1, 0):
            N.check(N.lib.hpc_rll_tune_set(3, mode))
            y, _ = m(x, None)
            g = torch.ones_like(y)
            y.backward(g, retain_graph=True)
            torch.cuda.synchronize()
            if os.environ.get("HPC_RLL_LSTM_PROFILE") == "1":
                continue
            for k, fn in (("fwd", lambda: m(x, None)), ("bwd", lambda: y.backward(g, retain_graph=True))):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                e1.synchronize()
                t = e0.elapsed_time(e1) / 5
                best[(mode, k)] = min(best.get((mode, k), 1e9), t)
        if os.environ.get("HPC_RLL_LSTM_PROFILE") == "1":
            break
    for mode in (1, 0):
        if (mode, "fwd") in best:
            print(f"S={S} B={B} I={I} H={H} L={L} persist={mode} stride={stride}: fwd {best[(mode, 'fwd')]:.3f} ms  "
                  f"bwd {best[(mode, 'bwd')]:.3f} ms", flush=True)
    N.check(N.lib.hpc_rll_tune_set(3, 1))
