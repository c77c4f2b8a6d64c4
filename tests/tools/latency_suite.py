#!/usr/bin/env python3
"""The reference's own comparison, at the reference's own test shapes (tests/test_*.py of DI-hpc): wall-clock per
forward+backward call (cuda.synchronize bracketed, like the *_perf() functions there) of the drop-in HIP modules vs
the same maths in eager PyTorch on the same GPU (the oracle evaluated on device, which is what `hpc_rll.origin` on
GPU does).  Small shapes: this measures launch + python overhead, not bandwidth."""
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
rows = []


def wall(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def rec(name, shape, t_hip, t_eager):
    r = dict(op=name, shape=shape, hip_us=t_hip * 1e6, eager_us=t_eager * 1e6, speedup=t_eager / t_hip)
    rows.append(r)
    print(json.dumps(r), flush=True)


def main():
    from hpc_rll.rl_utils.gae import GAE
    from hpc_rll.rl_utils.td import TDLambda, QNStepTD
    from hpc_rll.rl_utils.vtrace import VTrace
    from hpc_rll.rl_utils.upgo import UPGO
    from hpc_rll.rl_utils.ppo import PPO
    from hpc_rll.torch_utils.network.scatter_connection import ScatterConnection
    from hpc_rll.torch_utils.network.rnn import LSTM
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731

    T, B = 1024, 64
    v, r = rn(T + 1, B).requires_grad_(True), rn(T, B)
    m = GAE(T, B)
    rec("gae fwd", f"T={T} B={B}", wall(lambda: m(v, r)), wall(lambda: R.gae(v, r), 3))
    w = torch.rand(T, B, device=dev, generator=g)
    m2 = TDLambda(T, B)

    def f():
        v.grad = None
        m2(v, r, w).backward()

    def fo():
        v.grad = None
        R.td_lambda_error(v, r, w).backward()
    rec("td_lambda fwd+bwd", f"T={T} B={B}", wall(f), wall(fo, 3))

    T, B, N = 128, 128, 128
    to, bo = rn(T, B, N).requires_grad_(True), rn(T, B, N)
    a = torch.randint(0, N, (T, B), device=dev, generator=g)
    v, r = rn(T + 1, B).requires_grad_(True), rn(T, B)
    m3 = VTrace(T, B, N)

    def f():
        to.grad = None; v.grad = None
        sum(m3(to, bo, a, v, r)).backward()

    def fo():
        to.grad = None; v.grad = None
        sum(R.vtrace_error(to, bo, a, v, r)).backward()
    rec("vtrace fwd+bwd", f"T={T} B={B} N={N}", wall(f), wall(fo, 5))

    T, B, N = 256, 256, 256
    to = rn(T, B, N).requires_grad_(True)
    rho, a = rn(T, B), torch.randint(0, N, (T, B), device=dev, generator=g)
    r, v = rn(T, B), rn(T + 1, B)
    m4 = UPGO(T, B, N)

    def f():
        to.grad = None
        m4(to, rho, a, r, v).backward()

    def fo():
        to.grad = None
        R.upgo_loss(to, rho, a, r, v).backward()
    rec("upgo fwd+bwd", f"T={T} B={B} N={N}", wall(f), wall(fo, 5))

    B, N = 128, 128
    ln, lo = rn(B, N).requires_grad_(True), rn(B, N)
    a = torch.randint(0, N, (B,), device=dev, generator=g)
    vn, vo, adv, ret = rn(B).requires_grad_(True), rn(B), rn(B), rn(B)
    m5 = PPO(B, N)

    def f():
        ln.grad = None; vn.grad = None
        sum(m5(ln, lo, a, vn, vo, adv, ret)[0]).backward()

    def fo():
        ln.grad = None; vn.grad = None
        sum(R.ppo_error(ln, lo, a, vn, vo, adv, ret)[0]).backward()
    rec("ppo fwd+bwd", f"B={B} N={N}", wall(f), wall(fo, 10))

    T, B, N = 16, 64, 64
    q, nq = rn(B, N).requires_grad_(True), rn(B, N)
    a, na = torch.randint(0, N, (B,), device=dev, generator=g), torch.randint(0, N, (B,), device=dev, generator=g)
    r, done, w = rn(T, B), rn(B), rn(B)
    m6 = QNStepTD(T, B, N)

    def f():
        q.grad = None
        m6(q, nq, a, na, r, done, w, 0.95)[0].backward()

    def fo():
        q.grad = None
        R.q_nstep_td_error(q, nq, a, na, r, done, w, 0.95)[0].backward()
    rec("q_nstep_td fwd+bwd", f"T={T} B={B} N={N}", wall(f), wall(fo, 10))

    B, M, N, H, W = 256, 256, 256, 16, 16
    x = rn(B, M, N).requires_grad_(True)
    loc = torch.stack([torch.randint(0, H, (B, M), device=dev, generator=g), torch.randint(0, W, (B, M), device=dev, generator=g)], -1)
    m7 = ScatterConnection(B, M, N, H, W, "add")

    def f():
        x.grad = None
        (m7(x, loc) ** 2).mean().backward()

    def fo():
        x.grad = None
        cell = (loc[..., 0] * W + loc[..., 1]).unsqueeze(-1).expand(B, M, N)
        o = torch.zeros(B, H * W, N, device=dev).scatter_add(1, cell, x)
        (o ** 2).mean().backward()
    rec("scatter(add) fwd+bwd", f"B={B} M={M} N={N} H={H} W={W}", wall(f), wall(fo, 10))

    S, B, I, H, L = 64, 3, 1792, 384, 3
    m8 = LSTM(S, B, I, H, L).to(dev)
    x = rn(S, B, I).requires_grad_(True)

    def f():
        x.grad = None
        m8(x, None)[0].mean().backward()
    wx = [m8.wx[:I * 4 * H].reshape(I, 4 * H)] + [m8.wx[I * 4 * H + l * H * 4 * H: I * 4 * H + (l + 1) * H * 4 * H].reshape(H, 4 * H) for l in range(L - 1)]
    wh = [m8.wh[l * H * 4 * H:(l + 1) * H * 4 * H].reshape(H, 4 * H) for l in range(L)]
    z = torch.zeros(L, B, H, device=dev)

    def fo():
        x.grad = None
        R.lstm(x, z, z, wx, wh, m8.bias.reshape(L, 4 * H), m8.ln_gamma, m8.ln_beta)[0].mean().backward()
    rec("lstm fwd+bwd", f"S={S} B={B} I={I} H={H} L={L}", wall(f, 5), wall(fo, 2))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "latency_suite.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
