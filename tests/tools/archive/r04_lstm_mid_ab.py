#!/usr/bin/env python3
"""Mid-batch LSTM forward: the persistent mid-batch kernel (tune key 29 = 1, lstm_mid.hpp) against the two-launch step
(key 29 = 0), in one process on one box.  Per (B, H), I = H, L = 1, S = 64: microseconds per forward step (whole forward / S:
the x-branch product, its row statistics and the hn / cn copies included), no_grad and with a graph, and the same for a
narrow input (I = 64: the recurrence alone, to a few percent).  MODE=phases (with HPC_RLL_LSTM_PROFILE=1) runs each shape
once so that the kernel's own per-phase times are printed.  Writes gpurun_out/r04_lstm_mid_ab.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_torch_utils_network as N  # noqa: E402
from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402

dev = torch.device("cuda:0")
S = 64
phases = os.environ.get("MODE") == "phases"
if os.environ.get("REP"):
    N.tune_set(30, int(os.environ["REP"]))


def timed(fn, n=5, rounds=4):
    fn()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3


rows = []
shapes = [(B, H) for B in (16, 64, 256) for H in (384, 512, 1024)] + [(8, 1024), (32, 1024), (128, 1024), (128, 768)]
if phases:
    shapes = [(64, 1024), (64, 384), (256, 1024), (16, 512)]
for B, H in shapes:
    r = {"B": B, "H": H, "S": S}
    for I in ((H,) if phases else (H, 64)):
        torch.manual_seed(0)
        m = LSTM(S, B, I, H, 1).to(dev)
        x = torch.randn(S, B, I, device=dev)
        xg = x.clone().requires_grad_(True)
        h0, c0 = torch.randn(1, B, H, device=dev), torch.randn(1, B, H, device=dev)
        ys = {}
        for key in (0, 1, 2):
            N.tune_set(29, key)
            with torch.no_grad():
                y, _ = m(x, (h0, c0))
            torch.cuda.synchronize()
            path = N.lstm_last_forward_path()
            ys[key] = y
            if phases:
                continue

            def f_ng():
                with torch.no_grad():
                    m(x, (h0, c0))
            tag = f"key29_{key}" + ("" if I == H else "_I64")
            r[tag + "_path"] = path
            r[tag + "_nograd_us_per_step"] = timed(f_ng) / S
            r[tag + "_graph_us_per_step"] = timed(lambda: m(xg, (h0, c0))) / S
        N.tune_set(29, 2)
        r["max_abs_diff" + ("" if I == H else "_I64")] = float((ys[0] - ys[2]).abs().max())
    flops = 2.0 * B * 4 * H * H
    r["mfma_floor_us"] = flops / 157.3e12 * 1e6
    rows.append(r)
    print(json.dumps(r), flush=True)
if not phases:
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r04_lstm_mid_ab.json"), "w"), indent=1)
