#!/usr/bin/env python3
"""Mid-batch LSTM latency table: B = 16 ... 256, H = 384 ... 1024, I = H, L = 1, S = 64 -- the shapes an
RL actor / small learner runs.  Per (B, H): forward and backward microseconds per step, three readings each:
  eager      -- module(...) / .backward() as user code calls it (host launch path included);
  graph      -- the same launches replayed from ONE hipGraph (hpc_rll.graphed): the GPU-side time of the two-kernel step
                (product + cell) without the host;
  floor2     -- 2 * S dependent EMPTY launches replayed from a hipGraph: what the launch boundaries of a two-kernel-per-step
                design cost on this box before any work is done (the step cannot be faster than this without fusing kernels);
and the matrix-pipe / weight-stream lower bounds of the step itself: flops / (157.3 TFLOP/s) and the bytes of Wh read once
per step / 8 TB/s.  Writes gpurun_out/r04_lstm_mid_table.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_rll  # noqa: E402
from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402

dev = torch.device("cuda:0")
S = 64
TAG = ""
for _kv in os.environ.get("HPC_RLL_TUNE", "").split(","):      # e.g. HPC_RLL_TUNE=33:0 (A/B runs of this tool; tags the output file)
    if ":" in _kv:
        import hpc_torch_utils_network as _N
        _N.tune_set(int(_kv.split(":")[0]), int(_kv.split(":")[1]))
        TAG += "_" + _kv.replace(":", "-")


def timed(fn, n=5, rounds=3):
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
    return best * 1e3   # us


def floor2():
    """2*S dependent empty launches, as a graph replay (us per step)."""
    t = torch.zeros(64, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        t.add_(1.0)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(2 * S):
            t.add_(1.0)
    return timed(g.replay) / S


import faulthandler
faulthandler.enable()
rows, keep = [], []
fl2 = floor2()
for B in (16, 64, 256):
    for H in (384, 512, 1024):
        torch.manual_seed(0)
        m = LSTM(S, B, H, H, 1).to(dev)
        x = torch.randn(S, B, H, device=dev, requires_grad=True)
        h0, c0 = torch.zeros(1, B, H, device=dev), torch.zeros(1, B, H, device=dev)
        gy = torch.ones(S, B, H, device=dev)

        class Fwd(torch.nn.Module):
            def __init__(self, mod):
                super().__init__()
                self.mod = mod

            def forward(self, xx):
                return self.mod(xx, (h0, c0))[0]
        # GPU-side readings FIRST (before any eager backward has created AccumulateGrad nodes on the default stream): forward
        # alone and forward + backward as hipGraph replays
        gf = None
        try:
            fmod = Fwd(m)
            gfw = hpc_rll.graphed(fmod, x.detach(), backward=False)
            t_fg = timed(gfw.replay) / S
            gs = hpc_rll.graphed(fmod, x, grad_outputs=gy)
            t_fbg = timed(gs.replay) / S
            t_bg = max(t_fbg - t_fg, 0.0)
            keep.append((gfw, gs, fmod))      # (graphs stay alive until the end of the process)
        except Exception as e:  # noqa: BLE001
            t_fg = t_bg = None
            gf = repr(e)[:200]
        y, _ = m(x, (h0, c0))

        def bwd():
            x.grad = None
            for p in m.parameters():
                p.grad = None
            y.backward(gy, retain_graph=True)

        t_f = timed(lambda: m(x, (h0, c0))) / S
        t_b = timed(bwd) / S
        flops = 2.0 * B * 4 * H * H          # the recurrent product of one step
        r = {"B": B, "H": H, "S": S, "fwd_us_per_step": t_f, "bwd_us_per_step": t_b, "fwd_graph_us_per_step": t_fg,
             "bwd_graph_us_per_step": t_bg, "two_empty_launches_us_per_step": fl2,
             "mfma_floor_us": flops / 157.3e12 * 1e6, "wh_stream_floor_us": 4.0 * H * 4 * H / 8e12 * 1e6, "graph_error": gf}
        rows.append(r)
        print(json.dumps(r), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r04_lstm_mid_table" + TAG + ".json"), "w"), indent=1)
