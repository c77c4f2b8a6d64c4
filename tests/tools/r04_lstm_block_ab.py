#!/usr/bin/env python3
"""C4 LSTM (S=128, B=4096, I=H=1024, L=1): the persistent row-block forward (tune key 26 = 1) against one product + one cell
launch per step on the same gate-interleaved layout (key 26 = 0), and the start-skew knob (key 27), in ONE process on one
box.  Forward and backward ms (HIP events, median of 3 rounds x 2 calls), fraction of the fp32 MFMA peak.
Writes gpurun_out/r04_lstm_block_ab.json.  HPC_RLL_LSTM_PROFILE=1 prints the per-phase time of one workgroup."""
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


def timed(fn, n=2, rounds=3):
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


rows = []
configs = [tuple(int(v) for v in c.split(':')) for c in os.environ.get('CONFIGS', '0:0,1:10,9:10,25:10,9:0,9:16,1:10,0:0').split(',')]
for blk, skew in configs:
    N.tune_set(26, blk)
    N.tune_set(27, skew)
    y, _ = m(x, (h0, c0))
    path = N.lstm_last_forward_path()
    t_f = timed(lambda: m(x, (h0, c0)))
    g = torch.ones_like(y)

    def bwd():
        x.grad = None
        for p in m.parameters():
            p.grad = None
        y.backward(g, retain_graph=True)

    t_b = timed(bwd)
    torch.cuda.synchronize()
    assert N.async_error() == 0
    r = {"key26": blk, "key27_skew_us": skew, "path": path, "bwd_path": N.lstm_last_backward_path(), "fwd_ms": t_f, "fwd_frac": flops_f / (t_f * 1e-3) / PEAK,
         "bwd_ms": t_b, "bwd_frac": 2 * flops_f / (t_b * 1e-3) / PEAK, "y_checksum": float(y.double().sum().item())}
    rows.append(r)
    print(json.dumps(r), flush=True)
    del y, g
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r04_lstm_block_ab.json"), "w"), indent=1)
