#!/usr/bin/env python3
"""The GAE forward kernel reads 128.7 us in one process and 155.8 us in the next on the SAME box (the backward 112.8 both
times).  Is it the buffers' placement?  K independent buffer sets in one process, forward / backward alternating on each
set, kernel begin / end timestamps; virtual addresses printed beside the times."""
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi as N  # noqa: E402

lib = N.lib
dev = torch.device("cuda:0")
T, B = int(os.environ.get("PROBE_T", 1024)), int(os.environ.get("PROBE_B", 65536))
K = int(os.environ.get("PROBE_SETS", 8))
s = torch.cuda.current_stream().cuda_stream
coef = torch.empty(T, device=dev)
assert lib.hpc_rll_gae_coef(coef.data_ptr(), T, 0.99, 0.97, s) == 0
g = torch.Generator(device=dev).manual_seed(0)
pad = []
sets = []
for k in range(K):
    if os.environ.get("PROBE_JITTER") == "1":
        pad.append(torch.empty((k * 37 + 11) * 4096, device=dev))       # perturb the allocator's addresses
    v = torch.randn(T + 1, B, device=dev, generator=g)
    r = torch.randn(T, B, device=dev, generator=g)
    ga = torch.randn(T, B, device=dev, generator=g)
    sets.append((v, r, ga, torch.empty_like(r), torch.empty_like(v), torch.empty_like(r)))


def run(bs, n=40):
    v, r, ga, adv, gv, gr = bs
    f = lambda: lib.hpc_rll_gae_forward(v.data_ptr(), r.data_ptr(), adv.data_ptr(), coef.data_ptr(), T, B, 0.99, s)  # noqa: E731
    b = lambda: lib.hpc_rll_gae_backward(ga.data_ptr(), gv.data_ptr(), gr.data_ptr(), coef.data_ptr(), T, B, 0.99, s)  # noqa: E731
    for _ in range(5):
        f(); b()
    torch.cuda.synchronize()
    assert lib.hpc_rll_ktime_begin(2 * n) == 0
    for _ in range(n):
        f(); b()
    ms = (ctypes.c_float * (2 * n))()
    kd = (ctypes.c_int * (2 * n))()
    assert lib.hpc_rll_ktime_end(ms, kd, 2 * n) == 2 * n
    return statistics.median(ms[2 * i] for i in range(n)) * 1e3, statistics.median(ms[2 * i + 1] for i in range(n)) * 1e3


for _ in range(200):
    sets[0][3].add_(1.0)
for rnd in range(2):
    for k, bs in enumerate(sets):
        tf, tb = run(bs)
        print(f"round {rnd} set {k}: fwd {tf:6.1f} us  bwd {tb:6.1f} us | value {bs[0].data_ptr():#x} reward {bs[1].data_ptr():#x} adv {bs[3].data_ptr():#x} "
              f"| (reward-value)%2MiB = {(bs[1].data_ptr() - bs[0].data_ptr()) % (1 << 21):#x}", flush=True)
# mixed sets: value of set i with reward / adv of set j
v0, r0, ga0, a0, gv0, gr0 = sets[0]
for j in (1, 2, 3):
    mixed = (v0, sets[j][1], ga0, sets[j][3], gv0, gr0)
    tf, tb = run(mixed)
    print(f"value of set 0 with reward / adv of set {j}: fwd {tf:6.1f} us", flush=True)

# ---- which PAIR of arrays conflicts in a slow set, and does shifting one array by whole rows cure it?
times = {k: run(bs, 20)[0] for k, bs in enumerate(sets)}
slow = [k for k, t_ in times.items() if t_ > 145]
fast = [k for k, t_ in times.items() if t_ < 135]
print("slow sets", slow, "fast sets", fast, flush=True)
if slow and fast:
    ks, kf = slow[0], fast[0]
    vs, rs, gas, advs, gvs, grs = sets[ks]
    vf, rf, gaf, advf, gvf, grf = sets[kf]
    for name, bs in (("slow set, adv of a fast set", (vs, rs, gas, advf, gvs, grs)),
                     ("slow set, reward of a fast set", (vs, rf, gas, advs, gvs, grs)),
                     ("slow set, value of a fast set", (vf, rs, gas, advs, gvs, grs)),
                     ("slow set, reward AND adv of a fast set", (vs, rf, gas, advf, gvs, grs))):
        print(f"{name}: fwd {run(bs, 20)[0]:6.1f} us", flush=True)
    big = torch.empty((T + 40) * B, device=dev)
    for krows in (1, 2, 3, 4, 8, 16, 32):
        rk = big[krows * B:(krows + T) * B].view(T, B)
        rk.copy_(rs)
        print(f"slow set, reward moved into another allocation at +{krows} rows: fwd {run((vs, rk, gas, advs, gvs, grs), 20)[0]:6.1f} us "
              f"(the same view with a fast set's value: {run((vf, rk, gaf, advf, gvf, grf), 20)[0]:6.1f})", flush=True)
    big2 = torch.empty((T + 40) * B, device=dev)
    for krows in (0, 1, 2, 4, 8):
        ak = big2[krows * B:(krows + T) * B].view(T, B)
        print(f"slow set, adv moved into another allocation at +{krows} rows: fwd {run((vs, rs, gas, ak, gvs, grs), 20)[0]:6.1f} us", flush=True)

# ---- does the launch configuration matter on a slow set?
if slow and fast:
    def run_cfg(bs, c, n=20):
        v, r, ga, adv, gv, gr = bs
        f = lambda: lib.hpc_rll_gae_forward_ex(v.data_ptr(), r.data_ptr(), adv.data_ptr(), coef.data_ptr(), T, B, 0.99, *c, s)  # noqa: E731
        b = lambda: lib.hpc_rll_gae_backward(ga.data_ptr(), gv.data_ptr(), gr.data_ptr(), coef.data_ptr(), T, B, 0.99, s)  # noqa: E731
        if f() != 0:
            return float("nan")
        for _ in range(3):
            f(); b()
        torch.cuda.synchronize()
        assert lib.hpc_rll_ktime_begin(2 * n) == 0
        for _ in range(n):
            f(); b()
        ms = (ctypes.c_float * (2 * n))()
        kd = (ctypes.c_int * (2 * n))()
        assert lib.hpc_rll_ktime_end(ms, kd, 2 * n) == 2 * n
        return statistics.median(ms[2 * i] for i in range(n)) * 1e3
    for c in ((0, 0, 0, -1), (2, 4, 2, 11), (4, 4, 8, 11), (4, 8, 2, 11), (4, 4, 2, 11), (4, 8, 4, 11), (1, 8, 4, 11), (1, 16, 8, 11), (2, 8, 2, 3), (2, 8, 2, 1), (4, 8, 4, 3), (4, 4, 8, 3),
              (2, 16, 8, 11), (2, 8, 8, 11), (2, 4, 8, 11)):
        print(f"cfg {c}: slow set {run_cfg(sets[slow[0]], c):6.1f} us | fast set {run_cfg(sets[fast[0]], c):6.1f} us", flush=True)

# ---- robustness of candidate forward / backward configurations over ALL buffer sets (min / mean / max of the medians)
def run_cfg2(bs, cf, cb, n=16):
    v, r, ga, adv, gv, gr = bs
    f = lambda: lib.hpc_rll_gae_forward_ex(v.data_ptr(), r.data_ptr(), adv.data_ptr(), coef.data_ptr(), T, B, 0.99, *cf, s)  # noqa: E731
    b = lambda: lib.hpc_rll_gae_backward_ex(ga.data_ptr(), gv.data_ptr(), gr.data_ptr(), coef.data_ptr(), T, B, 0.99, *cb, s)  # noqa: E731
    if f() != 0 or b() != 0:
        return None
    for _ in range(3):
        f(); b()
    torch.cuda.synchronize()
    assert lib.hpc_rll_ktime_begin(2 * n) == 0
    for _ in range(n):
        f(); b()
    ms = (ctypes.c_float * (2 * n))()
    kd = (ctypes.c_int * (2 * n))()
    assert lib.hpc_rll_ktime_end(ms, kd, 2 * n) == 2 * n
    return statistics.median(ms[2 * i] for i in range(n)) * 1e3, statistics.median(ms[2 * i + 1] for i in range(n)) * 1e3


AUTO = (0, 0, 0, -1)
print("--- forward candidates over all sets: min / mean / max us", flush=True)
for c in (AUTO, (4, 4, 8, 43), (2, 4, 2, 11), (4, 4, 8, 11), (4, 8, 4, 11), (4, 4, 4, 11), (4, 8, 8, 11), (4, 8, 4, 3), (2, 8, 4, 11), (2, 4, 4, 11), (2, 16, 4, 11)):
    ts = [run_cfg2(bs, c, AUTO) for bs in sets]
    if ts[0] is None:
        continue
    f_ = [t_[0] for t_ in ts]
    print(f"fwd {c}: {min(f_):6.1f} / {sum(f_)/len(f_):6.1f} / {max(f_):6.1f}", flush=True)
print("--- backward candidates over all sets: min / mean / max us", flush=True)
for c in (AUTO, (4, 2, 4, 42), (4, 2, 4, 10), (4, 4, 2, 10), (2, 2, 4, 10), (2, 4, 2, 10), (2, 2, 4, 2), (4, 2, 8, 10)):
    ts = [run_cfg2(bs, AUTO, c) for bs in sets]
    if ts[0] is None:
        continue
    b_ = [t_[1] for t_ in ts]
    print(f"bwd {c}: {min(b_):6.1f} / {sum(b_)/len(b_):6.1f} / {max(b_):6.1f}", flush=True)
