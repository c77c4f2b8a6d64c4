#!/usr/bin/env python3
"""C4 LSTM backward: one row per workgroup + column-reduction pass (tune key 20 = 0) vs the row-walking cell that keeps
the bias / gamma / beta column sums (key 20 = workgroup count).  In-process, alternating rounds; the data gradients
must be bit-identical, the three parameter-gradient vectors equal to rounding.
Writes gpurun_out/r02_lstm_cellrows_probe.json."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi as N  # noqa: E402
from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


def run(S, B, I, H, L, knobs, rounds=3):
    torch.manual_seed(0)
    m = LSTM(S, B, I, H, L).to(dev)
    x = torch.randn(S, B, I, device=dev, requires_grad=True)
    y, _ = m(x, None)
    g = torch.randn_like(y)

    def bwd():
        x.grad = None
        for p in m.parameters():
            p.grad = None
        y.backward(g, retain_graph=True)

    res, grads = {}, {}
    for rnd in range(rounds + 1):
        for k in knobs:
            N.check(N.lib.hpc_rll_tune_set(20, k))
            if rnd == 0:
                bwd()
                grads[k] = {"x": x.grad.clone(), **{n: p.grad.clone() for n, p in m.named_parameters()}}
                continue
            res[k] = min(res.get(k, 1e9), timed(bwd))
    N.check(N.lib.hpc_rll_tune_set(20, 512))
    ref = grads[knobs[0]]
    diff = {}
    for k in knobs[1:]:
        diff[k] = {n: float((grads[k][n] - ref[n]).abs().max() / ref[n].abs().max()) for n in ref}
    row = {"shape": dict(S=S, B=B, I=I, H=H, L=L), "bwd_ms": res, "max_rel_diff_vs_first": diff}
    print(json.dumps(row), flush=True)
    return row


rows = [run(128, 4096, 1024, 1024, 1, [0, 256, 512]),
        run(32, 8192, 512, 512, 2, [0, 256, 512, 1024]),
        run(16, 8192, 1024, 1024, 1, [0, 256, 512, 1024])]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r02_lstm_cellrows_probe.json"), "w"), indent=1)
