#!/usr/bin/env python3
"""A/B of the large-batch TD-family forwards through the raw C ABI: the library before this round's changes to
csrc/dist_ops.hip (tests/tools/micro/libhpc_rll_hip_prev_td.so, built from commit a8043a5's source) against the current
one, alternating processes on ONE box (A, B, A, B) after a clock pre-roll.  Kernel time = HIP events around 20
back-to-back launches, median of 7.  Writes gpurun_out/r03_td_ab.json"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import cabi as N
    lib = N.lib
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(0)
    big = torch.empty(1 << 28, device=dev)
    for _ in range(300):      # clock pre-roll: ~0.3 s of streaming
        big.add_(1.0)
    if os.environ.get("AB_PREROLL") == "compute":      # ... or ~0.5 s of VALU-heavy work
        x = torch.randn(1 << 24, device=dev)
        for _ in range(40):
            for _ in range(20):
                x = torch.sin(x) * 1.0001
        del x
    torch.cuda.synchronize()
    import glob

    def sclk():
        vals = []
        for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
            for ln in open(f).read().splitlines():
                if ln.rstrip().endswith("*") and not ln.startswith("S"):
                    vals.append(float(ln.split(":")[1].strip().split("M")[0]))
        return max(vals) if vals else None
    B, Nq, nstep, n_atom, tau = 1 << 18, 64, 5, 51, 32
    reward = torch.randn(nstep, B, device=dev, generator=g)
    done = (torch.rand(B, device=dev, generator=g) < 0.1).float()
    weight = torch.rand(B, device=dev, generator=g)
    a = torch.randint(0, Nq, (B,), device=dev, generator=g)
    na = torch.randint(0, Nq, (B,), device=dev, generator=g)
    loss, td = torch.empty(1, device=dev), torch.empty(B, device=dev)
    part = torch.empty(int(lib.hpc_rll_partials_floats(B)), device=dev)
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
    d = torch.softmax(torch.randn(B, Nq, n_atom, device=dev, generator=g), -1)
    nd = torch.softmax(torch.randn(B, Nq, n_atom, device=dev, generator=g), -1)
    buf = torch.empty(B, n_atom, device=dev)
    out["c51_fwd_us"] = timed(lambda: lib.hpc_rll_dist_nstep_td_forward(P(d), P(nd), P(a), P(na), P(reward), P(done), P(weight), P(loss),
                                                                     P(td), P(buf), P(part), nstep, B, Nq, n_atom, 0.99, -10.0, 10.0, 1.0 / B, s))
    out["c51_loss"] = loss.item()
    out["sclk_after_c51"] = sclk()
    del d, nd
    q = torch.randn(B, Nq, tau, device=dev, generator=g)
    nq = torch.randn(B, Nq, tau, device=dev, generator=g)
    buf = torch.empty(B, tau, device=dev)
    out["qrdqn_fwd_us"] = timed(lambda: lib.hpc_rll_qrdqn_nstep_td_forward(P(q), P(nq), P(a), P(na), P(reward), P(done), P(weight), None, P(loss),
                                                                        P(td), P(buf), P(part), tau, nstep, B, Nq, 0.99, 1.0, 1.0 / B, s))
    out["qrdqn_loss"] = loss.item()
    out["sclk_after_qrdqn"] = sclk()
    del q, nq
    Bi = B // 4
    qi = torch.randn(tau, Bi, Nq, device=dev, generator=g)
    nqi = torch.randn(tau, Bi, Nq, device=dev, generator=g)
    rq = torch.rand(tau, Bi, device=dev, generator=g)
    ri, di, wi = reward[:, :Bi].contiguous(), done[:Bi].contiguous(), weight[:Bi].contiguous()
    ai, nai = a[:Bi].contiguous(), na[:Bi].contiguous()
    buf = torch.empty(Bi, tau, device=dev)
    out["iqn_fwd_us"] = timed(lambda: lib.hpc_rll_iqn_nstep_td_forward(P(qi), P(nqi), P(ai), P(nai), P(ri), P(di), P(rq), P(wi), None, P(loss),
                                                                    P(td), P(buf), P(part), tau, tau, nstep, Bi, Nq, 0.99, 1.0, 1.0 / Bi, s))
    out["iqn_loss"] = loss.item()
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    prev = os.path.join(ROOT, "tests", "tools", "micro", "libhpc_rll_hip_prev_td.so")
    res = {"prev": [], "now": []}
    for rnd in range(2):
        for tag, libp in (("prev", prev), ("now", None)):
            env = dict(os.environ)
            if rnd == 1:
                env["AB_PREROLL"] = "compute"
            if libp:
                env["HPC_RLL_LIB"] = libp
            else:
                env.pop("HPC_RLL_LIB", None)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True, env=env, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(tag, "failed", r.stdout[-500:], r.stderr[-1500:])
                continue
            res[tag].append(json.loads(line[0][7:]))
            print(tag, res[tag][-1])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r03_td_ab.json"), "w"), indent=1)
