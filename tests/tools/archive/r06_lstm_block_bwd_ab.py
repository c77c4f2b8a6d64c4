#!/usr/bin/env python3
"""Round 6 (key 26 bits 8-10 select lstm_block_bwd2_kernel<VAR>; 0 = the round-5 kernel).  Round 5 text: the C4 LSTM backward (S=128, B=4096, I=H=1024) through the variants of the persistent row-block BACKWARD kernel
(tune key 26 bits 8-10: two instead of three operand buffers / exchanges with cache-wide fences / four-row epilogue chunks),
in ONE process, interleaved over rounds.  Backward ms (HIP events), fraction of the fp32 matrix peak, gradient checksums (the
variants differ in nothing but scheduling: same bits expected).  HPC_RLL_LSTM_PROFILE=1 prints one workgroup's phase times.
Writes gpurun_out/r06_lstm_block_bwd_ab.json.
(the variants were compiled in at commit 9b6d159; only variant 5 = two buffers, fence-free exchanges, four-row chunks is left in the library)"""
import json
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
flops_f = 2.0 * S * B * 4 * H * (I + H) * L
PEAK = 157.3e12
VARS = [int(v) for v in os.environ.get("VARS", "0,2,1").split(",")]
ROUNDS = int(os.environ.get("ROUNDS", "2"))
NAMES = {0: "round-5 kernel", 1: "bwd2<1>: first chunk under the last k-tile", 2: "bwd2<0>: round-5 schedule in the new code"}

N.tune_set(26, 9)
y, _ = m(x, (h0, c0))
assert N.lstm_last_forward_path() == 4
g = torch.randn_like(y)
res = {v: [] for v in VARS}
sums = {}
for rnd in range(ROUNDS):
    for v in VARS:
        N.tune_set(26, 9 | (v << 8))

        def bwd():
            x.grad = None
            for p in m.parameters():
                p.grad = None
            y.backward(g, retain_graph=True)

        bwd()
        torch.cuda.synchronize()
        assert N.lstm_last_backward_path() == 4 and N.async_error() == 0
        sums.setdefault(v, (float(x.grad.double().sum()), float(m.wh.grad.double().sum()), float(m.ln_gamma.grad.double().sum()),
                            float(x.grad.double().abs().sum())))
        if rnd == 0:
            if v == VARS[0]:
                ref = (x.grad.clone(), m.wh.grad.clone(), m.ln_gamma.grad.clone(), m.wx.grad.clone())
            else:
                errs = [float((a_ - b_).abs().max() / b_.abs().max()) for a_, b_ in zip((x.grad, m.wh.grad, m.ln_gamma.grad, m.wx.grad), ref)]
                print(f"  var {v}: max |d| / max |ref| of dx, dWh, dgamma, dWx vs var {VARS[0]}: " + " ".join(f"{e_:.2e}" for e_ in errs), flush=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            bwd()
        e1.record()
        e1.synchronize()
        res[v].append(e0.elapsed_time(e1) / 3)
        print(f"round {rnd} var {v} ({NAMES[v]}): {res[v][-1]:.2f} ms", flush=True)
N.tune_set(26, 9)
rows = []
for v in VARS:
    t = statistics.median(res[v])
    rows.append({"variant": v, "what": NAMES[v], "bwd_ms": t, "all_ms": res[v], "bwd_frac": 2 * flops_f / (t * 1e-3) / PEAK, "checksums": sums[v]})
    print(json.dumps(rows[-1]))
same = all(sums[v] == sums[VARS[0]] for v in VARS)
print("all variants bit-identical (checksums):", same)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"rows": rows, "identical": same}, open(os.path.join(ROOT, "gpurun_out", "r06_lstm_block_bwd_ab.json"), "w"), indent=1)
