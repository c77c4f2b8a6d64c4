#!/usr/bin/env python3
"""Round 5: C4 LSTM backward with the row-block backward kernel on 128 rows x 128 units (one workgroup per CU, key 26 = 9) against
128 rows x 64 units (two workgroups per CU, key 26 = 25), one process, interleaved; gradient checksums must agree to rounding
(the row sums are combined from the same 32-unit partials in the same order: bit-identical expected).
HPC_RLL_LSTM_PROFILE=1 prints one workgroup's phase times.
(The 64-unit variant lives in commit 7e6db35 only; measured slower -- profiles/r05_lstm_bwd_bn_probe.txt -- and reverted.)"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_torch_utils_network as N  # noqa: E402
from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402

dev = torch.device("cuda:0")
S, B, I, H, L = (int(v) for v in os.environ.get("SHAPE", "128,4096,1024,1024,1").split(","))
torch.manual_seed(0)
m = LSTM(S, B, I, H, L).to(dev)
x = torch.randn(S, B, I, device=dev, requires_grad=True)
h0, c0 = torch.randn(L, B, H, device=dev), torch.randn(L, B, H, device=dev)
N.tune_set(26, 9)
y, _ = m(x, (h0, c0))
g = torch.randn_like(y)
keys = [int(k) for k in os.environ.get("KEYS", "9,25").split(",")]
res, sums = {k: [] for k in keys}, {}
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    for k in keys:
        N.tune_set(26, k)

        def bwd():
            x.grad = None
            for p in m.parameters():
                p.grad = None
            y.backward(g, retain_graph=True)
        bwd()
        torch.cuda.synchronize()
        assert N.lstm_last_backward_path() == 4 and N.async_error() == 0
        sums.setdefault(k, (float(x.grad.double().sum()), float(m.wh.grad.double().sum()), float(m.ln_gamma.grad.double().sum())))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            bwd()
        e1.record()
        e1.synchronize()
        res[k].append(e0.elapsed_time(e1) / 3)
N.tune_set(26, 9)
for k in keys:
    print(f"key26={k}: backward {statistics.median(res[k]):.2f} ms {['%.2f' % t for t in res[k]]} checksums {sums[k]}")
print("identical:", all(sums[k] == sums[keys[0]] for k in keys))
