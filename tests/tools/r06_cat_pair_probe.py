#!/usr/bin/env python3
"""Round 6: categorical forward with R rows per group in flight (HPC_RLL_CAT_R) and the V-trace head PAIR in one launch
(HPC_RLL_CAT_PAIR = rows per group, 0 = two launches), at C3 (T=256, B=16384, N=128) and N = 64 / 32, in one process,
interleaved rounds; results compared bit for bit with the default path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cabi as N_
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
lib = N_.lib

def t(fn, k=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / k * 1e3

from hpc_rll.rl_utils.vtrace import VTrace
from hpc_rll.rl_utils.gae import GAE
import time
def preroll(sec):
    """bench.py runs the suite after seconds of GAE work; the VALU-heavy categorical kernels read ~10 % slower in that state"""
    v = torch.randn(1025, 65536, device=dev, requires_grad=True); r = torch.randn(1024, 65536, device=dev, requires_grad=True)
    gg = torch.randn(1024, 65536, device=dev); m = GAE(1024, 65536)
    t0 = time.time()
    while time.time() - t0 < sec:
        for _ in range(200):
            v.grad = r.grad = None
            m(v, r).backward(gg)
        torch.cuda.synchronize()
preroll(float(os.environ.get("PREROLL_S", "4")))
from hpc_rll.rl_utils.upgo import UPGO
T, B = 256, 16384
for n in (128, 64, 32):
    g = torch.Generator(device=dev).manual_seed(0)
    rows = T * B
    xt = torch.randn(T, B, n, device=dev, generator=g); xb = torch.randn(T, B, n, device=dev, generator=g)
    a = torch.randint(0, n, (T, B), device=dev, generator=g)
    value = torch.randn(T + 1, B, device=dev, generator=g); reward = torch.randn(T, B, device=dev, generator=g)
    rho = torch.rand(T, B, device=dev, generator=g)
    logp, ent = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    vt, up = VTrace(T, B, n), UPGO(T, B, n)
    ref = {}
    res = {}
    for rnd in range(3):
        for r in (0, 4, 8):
            os.environ["HPC_RLL_CAT_R"] = str(r)
            fe = t(lambda: lib.hpc_rll_categorical_forward(xt.data_ptr(), a.data_ptr(), logp.data_ptr(), ent.data_ptr(), rows, n, st))
            if r == 0 and rnd == 0: ref["s"] = (logp.clone(), ent.clone())
            elif rnd == 0: assert torch.equal(logp, ref["s"][0]) and torch.equal(ent, ref["s"][1]), ("single", r)
            fn = t(lambda: lib.hpc_rll_categorical_forward(xt.data_ptr(), a.data_ptr(), logp.data_ptr(), None, rows, n, st))
            u = t(lambda: up(xt, rho, a, reward, value))
            k = res.setdefault(("R", r), [1e9] * 3)
            res[("R", r)] = [min(k[0], fe), min(k[1], fn), min(k[2], u)]
        os.environ["HPC_RLL_CAT_R"] = "0"
        for pr in (0, 1, 2, 4):
            os.environ["HPC_RLL_CAT_PAIR"] = str(pr)
            out = vt(xt, xb, a, value, reward)
            if pr == 0 and rnd == 0: ref["v"] = [o.clone() for o in out]
            elif rnd == 0: assert all(torch.equal(x, y) for x, y in zip(out, ref["v"])), ("pair", pr, [float(x) for x in out], [float(x) for x in ref["v"]])
            v = t(lambda: vt(xt, xb, a, value, reward))
            res[("PAIR", pr)] = [min(res.get(("PAIR", pr), [1e9])[0], v)]
        os.environ["HPC_RLL_CAT_PAIR"] = "0"
    print(f"N={n}: " + "  ".join(f"{k[0]}={k[1]}: " + "/".join(f"{x:.0f}" for x in v) for k, v in res.items()) + "  (R: fwd+ent / fwd / upgo fwd us;  PAIR: vtrace fwd us)", flush=True)
