#!/usr/bin/env python3
"""Tuning in the REAL access pattern: forward and backward launches ALTERNATE (as in training), so that no
launch finds its own inputs still sitting in the 256 MiB Infinity Cache from the previous identical launch.
Sweeps forward configurations against a fixed backward one, then backward ones against the best forward.
Per-kernel time = HIP events around every launch.  Writes gpurun_out/alt_gae_<T>x<B>.txt"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi as N  # noqa: E402

T = int(os.environ.get("TUNE_T", 1024))
B = int(os.environ.get("TUNE_B", 65536))
NREP = int(os.environ.get("NREP", 12))
lib = N.lib
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
v = torch.randn(T + 1, B, device=dev, generator=g)
r = torch.randn(T, B, device=dev, generator=g)
ga = torch.randn(T, B, device=dev, generator=g)
adv, gv, gr = torch.empty_like(r), torch.empty_like(v), torch.empty_like(r)
coef = torch.empty(T, device=dev)
s = torch.cuda.current_stream().cuda_stream
assert lib.hpc_rll_gae_coef(coef.data_ptr(), T, 0.99, 0.97, s) == 0
BYTES = 12 * T * B + 4 * B


def fwd(c):
    return lib.hpc_rll_gae_forward_ex(v.data_ptr(), r.data_ptr(), adv.data_ptr(), coef.data_ptr(), T, B, 0.99, *c, s)


def bwd(c):
    return lib.hpc_rll_gae_backward_ex(ga.data_ptr(), gv.data_ptr(), gr.data_ptr(), coef.data_ptr(), T, B, 0.99, *c, s)


def pair(cf, cb, n=NREP):
    if fwd(cf) != 0 or bwd(cb) != 0:
        return None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n + 1)]
    ev[0].record()
    for i in range(n):
        fwd(cf); ev[2 * i + 1].record()
        bwd(cb); ev[2 * i + 2].record()
    ev[-1].synchronize()
    tf = [ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(1, n)]
    tb = [ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(1, n)]
    return statistics.median(tf) * 1e-3, statistics.median(tb) * 1e-3


allc = [(vec, lc, nw, fl) for vec in (1, 2, 4) for lc in (2, 4, 8, 16) for nw in (2, 4, 8, 16) for fl in (0, 1, 2, 3)]
base_f, base_b = (0, 0, 0, -1), (0, 0, 0, -1)
lines = [f"T={T} B={B} bytes/launch={BYTES}  alternating fwd/bwd, median of {NREP-1}"]
rf = []
for c in allc:
    t = pair(c, base_b)
    if t:
        rf.append((t[0], t[1], c))
rf.sort()
lines.append("--- forward sweep (backward = auto): fwd_us GB/s | bwd_us | cfg(vec,lc,nw,flags)")
for tf, tb, c in rf[:30]:
    lines.append(f"{tf*1e6:7.1f} {BYTES/tf/1e9:6.0f} | {tb*1e6:7.1f} | {c}")
best_f = rf[0][2]
rb = []
for c in allc:
    t = pair(best_f, c)
    if t:
        rb.append((t[1], t[0], c))
rb.sort()
lines.append(f"--- backward sweep (forward = {best_f}): bwd_us GB/s | fwd_us | cfg")
for tb, tf, c in rb[:30]:
    lines.append(f"{tb*1e6:7.1f} {BYTES/tb/1e9:6.0f} | {tf*1e6:7.1f} | {c}")
t = pair(base_f, base_b)
lines.append(f"auto/auto: fwd {t[0]*1e6:.1f} us  bwd {t[1]*1e6:.1f} us")
t = pair(best_f, rb[0][2], 40)
lines.append(f"best/best {best_f} {rb[0][2]}: fwd {t[0]*1e6:.1f} us  bwd {t[1]*1e6:.1f} us  sum {sum(t)*1e6:.1f}")
txt = "\n".join(lines)
print(txt)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", f"alt_gae_{T}x{B}.txt"), "w").write(txt + "\n")

# ---- joint sweep on the sum of the pair (stores of one kernel are written back during the next one)
if os.environ.get("JOINT", "1") == "1":
    fc = [(2, 8, 2), (2, 4, 4), (2, 2, 8), (4, 8, 16), (4, 4, 16), (4, 8, 8), (4, 4, 8), (2, 16, 8), (1, 16, 16)]
    bc = [(2, 4, 4), (4, 4, 4), (4, 4, 16), (4, 2, 8), (4, 4, 8), (2, 8, 2), (2, 16, 8), (2, 2, 4), (1, 16, 16)]
    res = []
    for f in fc:
        for ff in (2, 3):
            for b in bc:
                for bf in (2, 3):
                    t = pair(f + (ff,), b + (bf,))
                    if t:
                        res.append((t[0] + t[1], t[0], t[1], f + (ff,), b + (bf,)))
    res.sort()
    jl = ["--- joint sweep sorted by fwd+bwd: sum | fwd | bwd | fwd cfg | bwd cfg"]
    for tot, tf, tb, f, b in res[:40]:
        jl.append(f"{tot*1e6:7.1f} | {tf*1e6:7.1f} | {tb*1e6:7.1f} | {f} | {b}")
    print("\n".join(jl))
    open(os.path.join(ROOT, "gpurun_out", f"alt_gae_{T}x{B}.txt"), "a").write("\n".join(jl) + "\n")
