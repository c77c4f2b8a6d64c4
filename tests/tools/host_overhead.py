#!/usr/bin/env python3
"""Where does the per-call time go at the reference's small test shapes?  cProfile of 300 fwd+bwd calls per op."""
import cProfile
import io
import os
import pstats
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731


def prof(name, fn, n=300, top=14):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / n
    t = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t) / n        # enqueue-only time (no sync): pure host cost if GPU keeps up
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        fn()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(top)
    print(f"==== {name}: wall {wall * 1e6:.1f} us/call, host-enqueue {host * 1e6:.1f} us/call")
    lines = s.getvalue().splitlines()
    print("\n".join(l[:150] for l in lines[6:6 + top + 2]))


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("ppo", "all"):
    from hpc_rll.rl_utils.ppo import PPO
    B, N = 128, 128
    ln, lo = rn(B, N).requires_grad_(True), rn(B, N)
    a = torch.randint(0, N, (B,), device=dev)
    vn, vo, adv, ret = rn(B).requires_grad_(True), rn(B), rn(B), rn(B)
    m = PPO(B, N)

    def f():
        ln.grad = None
        vn.grad = None
        loss, info = m(ln, lo, a, vn, vo, adv, ret)
        (loss.policy_loss + loss.value_loss + loss.entropy_loss).backward()
    prof("ppo fwd+bwd B=128 N=128", f)
if which in ("scatter", "all"):
    from hpc_rll.torch_utils.network.scatter_connection import ScatterConnection
    B, M, N, H, W = 256, 256, 256, 16, 16
    x = rn(B, M, N).requires_grad_(True)
    loc = torch.stack([torch.randint(0, H, (B, M), device=dev), torch.randint(0, W, (B, M), device=dev)], -1)
    m = ScatterConnection(B, M, N, H, W, "add")
    go = rn(B, N, H, W)

    def f2():
        x.grad = None
        m(x, loc).backward(go)
    prof("scatter(add) fwd+bwd", f2)
if which in ("q", "all"):
    from hpc_rll.rl_utils.td import QNStepTD
    T, B, N = 16, 64, 64
    q, nq = rn(B, N).requires_grad_(True), rn(B, N)
    a, na = torch.randint(0, N, (B,), device=dev), torch.randint(0, N, (B,), device=dev)
    r, d, w = rn(T, B), torch.zeros(B, device=dev), torch.ones(B, device=dev)
    m = QNStepTD(T, B, N)

    def f3():
        q.grad = None
        out = m(q, nq, a, na, r, d, w, 0.99)
        out[0].backward()
    prof("q_nstep_td fwd+bwd", f3)
if which in ("gae", "all"):
    from hpc_rll.rl_utils.gae import GAE
    T, B = 1024, 64
    v, r = rn(T + 1, B).requires_grad_(True), rn(T, B)
    m = GAE(T, B)
    ga = rn(T, B)

    def f4():
        v.grad = None
        m(v, r).backward(ga)
    prof("gae fwd+bwd T=1024 B=64", f4)
