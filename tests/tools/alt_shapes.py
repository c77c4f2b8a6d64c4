#!/usr/bin/env python3
"""Validate the GAE launch heuristic across shapes: for each (T,B) compare the auto configuration against the best of a
candidate set, in the alternating fwd/bwd pattern (see alt_gae.py).  Prints one line per shape."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cabi as N
lib = N.lib
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
CANDS_F = [(v, lc, nw, fl) for v in (1, 2, 4) for (lc, nw) in ((8, 2), (4, 4), (8, 4), (16, 8), (16, 16), (8, 8), (2, 8)) for fl in (2, 3) if not (v == 4 and lc == 16)]
CANDS_B = [(v, lc, nw, 2) for v in (1, 2, 4) for (lc, nw) in ((4, 4), (8, 2), (8, 4), (16, 8), (16, 16), (8, 8), (2, 8), (2, 4)) if not (v == 4 and lc == 16)]
for (T, B) in [(1024, 32768), (1024, 131072), (1024, 262144), (4096, 16384), (64, 1048576), (1024, 65536), (1024, 16384)]:
    g = torch.Generator(device=dev).manual_seed(0)
    v = torch.randn(T + 1, B, device=dev, generator=g); r = torch.randn(T, B, device=dev, generator=g); ga = torch.randn(T, B, device=dev, generator=g)
    adv, gv, gr = torch.empty_like(r), torch.empty_like(v), torch.empty_like(r)
    coef = torch.empty(T, device=dev); assert lib.hpc_rll_gae_coef(coef.data_ptr(), T, 0.99, 0.97, s) == 0
    fwd = lambda c: lib.hpc_rll_gae_forward_ex(v.data_ptr(), r.data_ptr(), adv.data_ptr(), coef.data_ptr(), T, B, 0.99, *c, s)
    bwd = lambda c: lib.hpc_rll_gae_backward_ex(ga.data_ptr(), gv.data_ptr(), gr.data_ptr(), coef.data_ptr(), T, B, 0.99, *c, s)
    def pair(cf, cb, n=10):
        if fwd(cf) != 0 or bwd(cb) != 0: return None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n + 1)]
        ev[0].record()
        for i in range(n):
            fwd(cf); ev[2 * i + 1].record(); bwd(cb); ev[2 * i + 2].record()
        ev[-1].synchronize()
        return (statistics.median(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(1, n)) * 1e-3,
                statistics.median(ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(1, n)) * 1e-3)
    auto = pair((0, 0, 0, -1), (0, 0, 0, -1))
    rf = sorted(((pair(c, (0, 0, 0, -1)) or (9, 9))[0], c) for c in CANDS_F)
    bf = rf[0]
    rb = sorted(((pair(bf[1], c) or (9, 9))[1], c) for c in CANDS_B)
    bb = rb[0]
    print("   fwd top:", " ".join(f"{t*1e6:.1f}{c}" for t, c in rf[:6]))
    print("   bwd top:", " ".join(f"{t*1e6:.1f}{c}" for t, c in rb[:6]))
    best = pair(bf[1], bb[1])
    by = 12 * T * B + 4 * B
    print(f"T={T:5d} B={B:8d} {by/1e6:8.1f} MB | auto fwd {auto[0]*1e6:7.1f} bwd {auto[1]*1e6:7.1f} us ({2*by/sum(auto)/1e9:5.0f} GB/s) | "
          f"best fwd {best[0]*1e6:7.1f} {bf[1]} bwd {best[1]*1e6:7.1f} {bb[1]} ({2*by/sum(best)/1e9:5.0f} GB/s)", flush=True)
    del v, r, ga, adv, gv, gr
