#!/usr/bin/env python3
"""Round-2 host-path targets: API-level wall time per call of the drop-in modules now that the
L2 is a compiled torch extension (C++ autograd nodes) instead of ctypes + python autograd.Function.

  * GAE fwd+bwd at T=1024,B=64 (the reference's test shape)            target <= 30 us
  * TD-lambda backward at C3 (T=256,B=16384)                           target <= 15 us
  * ScatterConnection(add) fwd+bwd at the reference test shape vs eager PyTorch   target >= 3x
  * GAE fwd+bwd eager at B=8192 per rank (8-GPU strong scaling of B=65536)        target <= 60 us / step
Prints one JSON line per row; `python tests/tools/r02_latency_targets.py > gpurun_out/r02_latency.json`."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import ref_torch as R  # noqa: E402  (comparison baseline only)

dev = torch.device("cuda:0")


def wall(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / n)
    return best * 1e6


def host_only(fn, n=200):
    """Host time to ISSUE the call (no synchronise inside the loop, the queue is drained before and after)."""
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    return dt * 1e6


def out(**kw):
    print(json.dumps(kw), flush=True)


def main():
    from hpc_rll.rl_utils.gae import GAE
    from hpc_rll.rl_utils.td import TDLambda, QNStepTD
    from hpc_rll.rl_utils.ppo import PPO
    from hpc_rll.rl_utils.vtrace import VTrace
    from hpc_rll.torch_utils.network.scatter_connection import ScatterConnection
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731

    # ---- the floor: torch's own smallest fwd+bwd through the autograd engine on this box (one mul kernel each way)
    x0 = rn(64).requires_grad_(True)
    g0 = rn(64)

    def fb0():
        x0.grad = None
        (x0 * 2.0).backward(g0)
    out(op="torch reference: (x*2).backward(g)", shape="64", wall_us=wall(fb0), issue_us=host_only(fb0))

    # ---- GAE at the reference test shape and at the strong-scaling per-rank shape
    for T, B, target in ((1024, 64, 30.0), (1024, 8192, 60.0), (1024, 1024, None)):
        v, r, ga = rn(T + 1, B).requires_grad_(True), rn(T, B).requires_grad_(True), rn(T, B)
        m = GAE(T, B)

        def fb():
            v.grad = None
            r.grad = None
            m(v, r).backward(ga)

        def fwd():
            with torch.no_grad():
                m(v, r)
        out(op="gae fwd+bwd", shape=f"T={T} B={B}", wall_us=wall(fb), issue_us=host_only(fb), fwd_only_us=wall(fwd),
            target_us=target)

    # ---- TD-lambda at C3: fwd+bwd minus fwd = backward
    T, B = 256, 16384
    v, r, w = rn(T + 1, B).requires_grad_(True), rn(T, B), torch.rand(T, B, device=dev, generator=g)
    m2 = TDLambda(T, B)

    def fb():
        v.grad = None
        m2(v, r, w).backward()

    def fwd():
        with torch.no_grad():
            m2(v, r, w)
    loss = m2(v, r, w)

    def bwd():
        v.grad = None
        loss.backward(retain_graph=True)
    out(op="td_lambda", shape=f"T={T} B={B}", fwd_bwd_us=wall(fb), fwd_us=wall(fwd), bwd_us=wall(bwd),
        bwd_issue_us=host_only(bwd), target_bwd_us=15.0)
    T, B = 1024, 64
    v, r, w = rn(T + 1, B).requires_grad_(True), rn(T, B), torch.rand(T, B, device=dev, generator=g)
    m2 = TDLambda(T, B)
    out(op="td_lambda", shape=f"T={T} B={B}", fwd_bwd_us=wall(fb), fwd_us=wall(fwd))

    # ---- ScatterConnection at the reference test shape (tests/test_scatter_connection.py: B=256,M=256,N=256,H=W=16)
    B, M, N, H, W = 256, 256, 256, 16, 16
    x = rn(B, M, N).requires_grad_(True)
    loc = torch.stack([torch.randint(0, H, (B, M), device=dev, generator=g),
                       torch.randint(0, W, (B, M), device=dev, generator=g)], -1)
    go = rn(B, N, H, W)
    for typ in ("add", "cover"):
        ms = ScatterConnection(B, M, N, H, W, typ)

        def fb():
            x.grad = None
            ms(x, loc).backward(go)

        def fo():
            x.grad = None
            R.scatter_connection(x, loc, H, W, typ).backward(go)
        t_hip, t_eager = wall(fb, 50), wall(fo, 20, 3)
        out(op=f"scatter({typ}) fwd+bwd", shape=f"B={B} M={M} N={N} {H}x{W}", hip_us=t_hip, eager_us=t_eager,
            speedup=t_eager / t_hip, target_speedup=3.0)

    # ---- small per-sample ops (host-bound)
    B, N = 128, 128
    ln, lo = rn(B, N).requires_grad_(True), rn(B, N)
    a = torch.randint(0, N, (B,), device=dev, generator=g)
    vn, vo, adv, ret = rn(B).requires_grad_(True), rn(B), rn(B), rn(B)
    m5 = PPO(B, N)

    def fb():
        ln.grad = None
        vn.grad = None
        sum(m5(ln, lo, a, vn, vo, adv, ret)[0]).backward()
    out(op="ppo fwd+bwd", shape=f"B={B} N={N}", wall_us=wall(fb), r01_us=205.0)

    T, B, N = 16, 64, 64
    q, nq = rn(B, N).requires_grad_(True), rn(B, N)
    a, na = torch.randint(0, N, (B,), device=dev, generator=g), torch.randint(0, N, (B,), device=dev, generator=g)
    r, done, w = rn(T, B), rn(B), rn(B)
    m6 = QNStepTD(T, B, N)

    def fb():
        q.grad = None
        m6(q, nq, a, na, r, done, w, 0.95)[0].backward()
    out(op="q_nstep_td fwd+bwd", shape=f"nstep={T} B={B} N={N}", wall_us=wall(fb), r01_us=129.0)

    T, B, N = 128, 128, 128
    to, bo = rn(T, B, N).requires_grad_(True), rn(T, B, N)
    a = torch.randint(0, N, (T, B), device=dev, generator=g)
    v, r = rn(T + 1, B).requires_grad_(True), rn(T, B)
    m3 = VTrace(T, B, N)

    def fb():
        to.grad = None
        v.grad = None
        sum(m3(to, bo, a, v, r)).backward()
    out(op="vtrace fwd+bwd", shape=f"T={T} B={B} N={N}", wall_us=wall(fb))

    # ---- LAST (toggling the engine mode changes later timings): the autograd engine's cross-thread hand-off.  With
    # multithreading off backward runs on the calling thread; the difference is torch's, not this library's.
    out(op="torch reference again", shape="64", wall_us=wall(fb0))
    T, B = 1024, 64
    v, r, ga = rn(T + 1, B).requires_grad_(True), rn(T, B).requires_grad_(True), rn(T, B)
    m = GAE(T, B)

    def fbg():
        v.grad = None
        r.grad = None
        m(v, r).backward(ga)
    t_on = wall(fbg)
    torch.autograd.set_multithreading_enabled(False)
    out(op="gae fwd+bwd, autograd multithreading off", shape=f"T={T} B={B}", wall_us=wall(fbg), multithreaded_us=t_on)
    out(op="torch reference, autograd multithreading off", shape="64", wall_us=wall(fb0))
    torch.autograd.set_multithreading_enabled(True)


if __name__ == "__main__":
    main()
