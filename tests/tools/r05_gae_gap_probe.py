#!/usr/bin/env python3
"""Round 5: where does a rank of the 8-GPU strong-scaling run (GAE, T = 1024, B = 8192) spend its step?

    python tests/tools/r05_gae_gap_probe.py            # step time by launch path
    MODE=trace python tests/tools/r05_gae_gap_probe.py # short replay loops for rocprofv3 --kernel-trace
    MODE=gaps python tests/tools/r05_gae_gap_probe.py <results.db>   # kernel durations and the GPU-side gaps between them

Launch paths, same kernels each: (a) the C ABI ops called directly, forward then backward, eager; (b) the drop-in module +
autograd, eager (bench.py's eager leg); (c) hpc_rll.graphed: one hipGraphLaunch per step; (d) n steps -- n micro-batches, each with
its OWN input and gradient buffers -- captured into one graph (hpc_rll.graphed_steps): one hipGraphLaunch per n steps."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
MODE = os.environ.get("MODE", "time")


def gaps(db):
    import re
    import sqlite3
    import statistics as st
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table' or type='view'")]
    ktab = [t for t in tabs if t == "kernels"] or [t for t in tabs if t.startswith("kernels")]
    rows = con.execute(f"select name, start, end from {ktab[0]} order by start").fetchall()

    def short(n):
        m = re.search(r"(gae_\w+|\w+_kernel\w*|__amd_rocclr_\w+)", n)
        return m.group(1) if m else n[:40]
    ks = [(short(n), s, e) for n, s, e in rows]
    print(f"# {len(ks)} kernel launches in the trace; by name:", {n: sum(1 for k in ks if k[0] == n) for n in sorted({k[0] for k in ks})})
    # the three timed loops are separated by synchronisations (idle > 200 us): segment, then per segment durations and gaps
    seg, segs = [], []
    for i, k in enumerate(ks):
        if seg and k[1] - seg[-1][2] > 200e3:
            segs.append(seg)
            seg = []
        seg.append(k)
    segs.append(seg)
    for sg in segs:
        if len(sg) < 100:
            continue
        dur, gap = {}, {}
        for i, (n, s, e) in enumerate(sg):
            dur.setdefault(n, []).append((e - s) / 1e3)
            if i:
                gap.setdefault(sg[i - 1][0][:11] + " -> " + n[:11], []).append((s - sg[i - 1][2]) / 1e3)
        span = (sg[-1][2] - sg[0][1]) / 1e3
        nf = sum(1 for k in sg if "fwd" in k[0])
        print(f"## segment of {len(sg)} launches, {span / max(nf, 1):.2f} us per forward launch (first start .. last end)")
        for n, d in dur.items():
            print(f"   kernel {n:26s} n={len(d):5d} median {st.median(d):6.2f} us  mean {st.mean(d):6.2f}  min {min(d):6.2f}")
        for k, g in gap.items():
            g2 = sorted(g)
            print(f"   gap {k:26s} n={len(g):5d} median {st.median(g):6.2f} us  p10 {g2[len(g2) // 10]:6.2f}  p90 {g2[9 * len(g2) // 10]:6.2f}  mean {st.mean(g):6.2f}")


if MODE == "gaps":
    gaps(sys.argv[1])
    sys.exit(0)

import torch  # noqa: E402

import hpc_rl_utils as U  # noqa: E402
import hpc_rll  # noqa: E402
from hpc_rll.rl_utils.gae import GAE  # noqa: E402

dev = torch.device("cuda:0")
T, B = 1024, int(os.environ.get("B", "8192"))
gamma, lam = 0.99, 0.97


def bufs(seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    v = torch.randn(T + 1, B, device=dev, generator=g).requires_grad_(True)
    r = torch.randn(T, B, device=dev, generator=g).requires_grad_(True)
    ga = torch.randn(T, B, device=dev, generator=g)
    return v, r, ga


def timed(step, steps, reps=3, per=1):
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / (steps * per) * 1e6)
    return sorted(out)


v, r, ga = bufs(1)
gae = GAE(T, B).to(dev)
vd, rd = v.detach(), r.detach()
adv, gv, gr = torch.empty_like(rd), torch.empty_like(vd), torch.empty_like(rd)


def direct():
    U.GaeForward([vd, rd], [adv], gamma, lam)
    U.GaeBackward([ga], [gv, gr], gamma, lam)


def eager():
    v.grad = None
    r.grad = None
    gae(v, r, gamma, lam).backward(ga)


g1 = hpc_rll.graphed(gae, v, r, gamma, lam, grad_outputs=ga)
multi, same = {}, {}
for n in (2, 4, 8):
    sets = [bufs(100 + i) for i in range(n)]
    multi[n] = (hpc_rll.graphed_steps(gae, [(s[0], s[1], gamma, lam) for s in sets], grad_outputs=[s[2] for s in sets]), sets)
    # the SAME buffers n times (what the eager loops above do as well: every step re-reads the one synthetic batch)
    same[n] = hpc_rll.graphed_steps(gae, [(v, r, gamma, lam)] * n, grad_outputs=[ga] * n)

if MODE == "trace":
    for fn, cnt in ((g1.replay, 300), (same[4].replay, 80), (direct, 300), (eager, 300)):
        for _ in range(cnt):
            fn()
        torch.cuda.synchronize()
        time.sleep(0.01)
    sys.exit(0)

print(f"# GAE fwd+bwd, T={T} B={B}: microseconds per step (sorted rounds of 300 steps)")
print("direct C-ABI ops, eager      ", ["%.2f" % t for t in timed(direct, 300)])
print("module + autograd, eager     ", ["%.2f" % t for t in timed(eager, 300)])
print("hpc_rll.graphed, 1 step/graph", ["%.2f" % t for t in timed(g1.replay, 300)])
for n, (gs, _) in multi.items():
    print(f"graphed_steps, {n} steps/graph, n buffer sets ", ["%.2f" % t for t in timed(gs.replay, 300 // n, per=n)])
for n, gs in same.items():
    print(f"graphed_steps, {n} steps/graph, same buffers  ", ["%.2f" % t for t in timed(gs.replay, 300 // n, per=n)])
# bit-identity of the multi-step graph with the single-step one on the same inputs
gs, sets = multi[4]
gs.replay()
torch.cuda.synchronize()
ok = True
for i, s in enumerate(sets):
    a = gae(s[0], s[1], gamma, lam)
    gvi, gri = torch.autograd.grad(a, (s[0], s[1]), s[2])
    ok = ok and torch.equal(a, gs.outputs[i]) and torch.equal(gvi, gs.grads[i][0]) and torch.equal(gri, gs.grads[i][1])
print("graphed_steps results identical to eager:", ok)
