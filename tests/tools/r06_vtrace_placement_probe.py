#!/usr/bin/env python3
"""Round 6: is the V-trace / categorical forward's box-to-box spread (0.645 vs 0.74 ms) a placement effect?  K independent sets of
(target, behaviour) logits in one process, module forward and the bare categorical forward on each; then pairs mixed."""
import os, sys, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cabi as C
from hpc_rll.rl_utils.vtrace import VTrace
from hpc_rll.rl_utils.gae import GAE
dev = torch.device("cuda:0")
def preroll(sec):
    v = torch.randn(1025, 65536, device=dev, requires_grad=True); r = torch.randn(1024, 65536, device=dev, requires_grad=True)
    gg = torch.randn(1024, 65536, device=dev); m = GAE(1024, 65536)
    t0 = time.time()
    while time.time() - t0 < sec:
        for _ in range(200):
            v.grad = r.grad = None
            m(v, r).backward(gg)
        torch.cuda.synchronize()
preroll(3)
def t(fn, k=5, rounds=3):
    fn(); ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / k * 1e3)
    return statistics.median(ts)
T, B, N = 256, 16384, 128
K = int(os.environ.get("PROBE_SETS", 8))
g = torch.Generator(device=dev).manual_seed(0)
value = torch.randn(T + 1, B, device=dev, generator=g); reward = torch.randn(T, B, device=dev, generator=g)
a = torch.randint(0, N, (T, B), device=dev, generator=g)
sets, pad = [], []
for k in range(K):
    pad.append(torch.empty((k * 53 + 7) * 4096, device=dev))
    sets.append((torch.randn(T, B, N, device=dev, generator=g), torch.randn(T, B, N, device=dev, generator=g)))
vt = VTrace(T, B, N)
rows = T * B
lp, en = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
for rnd in range(2):
    out = []
    for k, (xt, xb) in enumerate(sets):
        os.environ["HPC_RLL_CAT_R"] = "0"
        tv = t(lambda: vt(xt, xb, a, value, reward))
        tc = t(lambda: C.call("hpc_rll_categorical_forward", dev, xt.data_ptr(), a.data_ptr(), lp.data_ptr(), en.data_ptr(), rows, N))
        tb = t(lambda: C.call("hpc_rll_categorical_forward", dev, xb.data_ptr(), a.data_ptr(), lp.data_ptr(), 0, rows, N))
        os.environ["HPC_RLL_CAT_R"] = "4"
        t4 = t(lambda: C.call("hpc_rll_categorical_forward", dev, xt.data_ptr(), a.data_ptr(), lp.data_ptr(), en.data_ptr(), rows, N))
        os.environ["HPC_RLL_CAT_R"] = "8"
        t8 = t(lambda: C.call("hpc_rll_categorical_forward", dev, xt.data_ptr(), a.data_ptr(), lp.data_ptr(), en.data_ptr(), rows, N))
        os.environ["HPC_RLL_CAT_R"] = "0"
        out.append((tv, tc, tb, t4, t8))
    print(f"round {rnd}: vtrace fwd / categorical(target)+ent / categorical(behaviour) us per set: " + "  ".join(f"{x[0]:.0f}/{x[1]:.0f}/{x[2]:.0f}/R4:{x[3]:.0f}/R8:{x[4]:.0f}" for x in out), flush=True)
