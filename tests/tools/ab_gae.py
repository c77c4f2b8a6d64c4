#!/usr/bin/env python3
"""Interleaved A/B of GAE kernel configurations (and experimental library builds) at one shape.
Usage: ab_gae.py [lib1.so lib2.so ...]   (default: the product library).  Writes gpurun_out/ab_gae_<T>x<B>.txt"""
import ctypes
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
ROUNDS = int(os.environ.get("ROUNDS", 7))
libs = {"base": N.lib}
for p in sys.argv[1:]:
    l = ctypes.CDLL(p)
    for name, args in N.SIGNATURES.items():
        if hasattr(l, name):
            getattr(l, name).argtypes = args
            getattr(l, name).restype = ctypes.c_int
    libs[os.path.basename(p).replace(".so", "")] = l
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
v = torch.randn(T + 1, B, device=dev, generator=g)
r = torch.randn(T, B, device=dev, generator=g)
ga = torch.randn(T, B, device=dev, generator=g)
adv, gv, gr = torch.empty_like(r), torch.empty_like(v), torch.empty_like(r)
coef = torch.empty(T, device=dev)
s = torch.cuda.current_stream().cuda_stream
assert N.lib.hpc_rll_gae_coef(coef.data_ptr(), T, 0.99, 0.97, s) == 0
BYTES = 12 * T * B + 4 * B
cfgs = [tuple(int(x) for x in c.split(",")) for c in os.environ.get(
    "CFGS", "4,2,4;4,2,8;4,2,16;4,4,2;4,4,4;4,4,8;4,8,2;4,8,4;2,8,2;2,4,4;2,2,8;2,4,8;0,0,0").split(";")]


def timed(fn, n=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


res = {}
for rnd in range(ROUNDS):
    for ln, lib in libs.items():
        for (vec, lc, nw) in cfgs:
            f = lambda: lib.hpc_rll_gae_forward_ex(v.data_ptr(), r.data_ptr(), adv.data_ptr(), coef.data_ptr(), T, B, 0.99, vec, lc, nw, s)
            b = lambda: lib.hpc_rll_gae_backward_ex(ga.data_ptr(), gv.data_ptr(), gr.data_ptr(), coef.data_ptr(), T, B, 0.99, vec, lc, nw, s)
            if f() != 0 or b() != 0:
                continue
            res.setdefault((ln, vec, lc, nw), []).append((timed(f), timed(b)))
lines = [f"T={T} B={B} bytes/launch={BYTES} rounds={ROUNDS}",
         "lib          vec lc nw | fwd med  min (us)  GB/s(med) | bwd med  min (us)  GB/s(med) | sum med"]
rows = []
for k, ts in res.items():
    fm, fmin = statistics.median(t[0] for t in ts), min(t[0] for t in ts)
    bm, bmin = statistics.median(t[1] for t in ts), min(t[1] for t in ts)
    rows.append((fm + bm, k, fm, fmin, bm, bmin))
for tot, k, fm, fmin, bm, bmin in sorted(rows):
    lines.append(f"{k[0]:12s} {k[1]:3d} {k[2]:2d} {k[3]:2d} | {fm*1e6:7.1f} {fmin*1e6:7.1f} {BYTES/fm/1e9:8.0f} | "
                 f"{bm*1e6:7.1f} {bmin*1e6:7.1f} {BYTES/bm/1e9:8.0f} | {tot*1e6:7.1f}")
txt = "\n".join(lines)
print(txt)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", f"ab_gae_{T}x{B}.txt"), "w").write(txt + "\n")
