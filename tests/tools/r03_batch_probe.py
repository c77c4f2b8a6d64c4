#!/usr/bin/env python3
"""Large-batch C51 / QR-DQN forwards: the SW-samples-per-wave kernels (tune key 24 = 16 / 32 / 64) against the previous
kernels (key 24 = 1) in ONE process, through the raw C ABI.  Kernel time = HIP events around 20 back-to-back launches,
median of 7, after a clock pre-roll; results compared with the key-24 = 1 outputs.  Writes gpurun_out/r03_batch_probe.json"""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi as N  # noqa: E402

lib = N.lib
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
big = torch.empty(1 << 28, device=dev)
for _ in range(300):
    big.add_(1.0)
del big
torch.cuda.synchronize()
P = lambda t: t.data_ptr()  # noqa: E731


def timed(fn, n=20, rounds=7):
    assert fn() == 0
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)


out = {}
for B in [int(x) for x in os.environ.get('PROBE_B', '262144,131072,65536').split(',')]:
    Nq, nstep, n_atom, tau = 64, 5, 51, 32
    reward = torch.randn(nstep, B, device=dev, generator=g)
    done = (torch.rand(B, device=dev, generator=g) < 0.1).float()
    weight = torch.rand(B, device=dev, generator=g)
    a = torch.randint(0, Nq, (B,), device=dev, generator=g)
    na = torch.randint(0, Nq, (B,), device=dev, generator=g)
    loss, td = torch.empty(1, device=dev), torch.empty(B, device=dev)
    part = torch.empty(int(lib.hpc_rll_partials_floats(B)), device=dev)
    d = torch.softmax(torch.randn(B, Nq, n_atom, device=dev, generator=g), -1)
    nd = torch.softmax(torch.randn(B, Nq, n_atom, device=dev, generator=g), -1)
    buf = torch.empty(B, n_atom, device=dev)
    ref = None
    for sw in [int(x) for x in os.environ.get('PROBE_SW', '1,16,32,64').split(',')]:
        assert lib.hpc_rll_tune_set(24, sw) == 0
        td.zero_(); buf.zero_()
        us = timed(lambda: lib.hpc_rll_dist_nstep_td_forward(P(d), P(nd), P(a), P(na), P(reward), P(done), P(weight), P(loss),
                                                          P(td), P(buf), P(part), nstep, B, Nq, n_atom, 0.99, -10.0, 10.0, 1.0 / B, s))
        cur = (td.clone(), buf.clone(), loss.item())
        if ref is None:
            ref = cur
        out[f"c51_B{B}_sw{sw}"] = {"us": us, "loss": cur[2], "td_maxabs_vs_sw1": (cur[0] - ref[0]).abs().max().item(),
                                    "td_max": ref[0].abs().max().item(),
                                    "buf_maxabs_vs_sw1": (cur[1] - ref[1]).abs().max().item(), "buf_max": ref[1].abs().max().item()}
        print(f"c51 B={B} sw={sw}", out[f"c51_B{B}_sw{sw}"], flush=True)
    del d, nd
    q = torch.randn(B, Nq, tau, device=dev, generator=g)
    nq = torch.randn(B, Nq, tau, device=dev, generator=g)
    buf = torch.empty(B, tau, device=dev)
    ref = None
    for sw in [int(x) for x in os.environ.get('PROBE_SW', '1,16,32,64').split(',')]:
        assert lib.hpc_rll_tune_set(24, sw) == 0
        td.zero_(); buf.zero_()
        us = timed(lambda: lib.hpc_rll_qrdqn_nstep_td_forward(P(q), P(nq), P(a), P(na), P(reward), P(done), P(weight), None, P(loss),
                                                           P(td), P(buf), P(part), tau, nstep, B, Nq, 0.99, 1.0, 1.0 / B, s))
        cur = (td.clone(), buf.clone(), loss.item())
        if ref is None:
            ref = cur
        out[f"qrdqn_B{B}_sw{sw}"] = {"us": us, "loss": cur[2], "td_bit_identical": bool((cur[0] == ref[0]).all().item()),
                                      "td_maxabs_vs_sw1": (cur[0] - ref[0]).abs().max().item(),
                                      "buf_bit_identical": bool((cur[1] == ref[1]).all().item())}
        print(f"qrdqn B={B} sw={sw}", out[f"qrdqn_B{B}_sw{sw}"], flush=True)
    del q, nq
lib.hpc_rll_tune_set(24, 0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_batch_probe.json"), "w"), indent=1)
