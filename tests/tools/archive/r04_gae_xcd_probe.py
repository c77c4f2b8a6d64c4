#!/usr/bin/env python3
"""GAE at T=1024, B=65536: the shipped kernels against the same kernels with XCD-contiguous column tiles (flags bit 5): with the
identity mapping the four 1 KiB pieces of every 4 KiB-aligned block of a row are written by workgroups on four different XCDs
(profiles/r04_writebw.txt: a pure-write stream wants whole 4 KiB blocks from one CU).  Alternating fwd / bwd, kernel begin / end
timestamps, three buffer sets."""
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi as N  # noqa: E402

T, B = 1024, 65536
lib = N.lib
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
coef = torch.empty(T, device=dev)
s = torch.cuda.current_stream().cuda_stream
assert lib.hpc_rll_gae_coef(coef.data_ptr(), T, 0.99, 0.97, s) == 0
BYTES = 12 * T * B + 4 * B
AUTO = (0, 0, 0, -1)


def bufs():
    v = torch.randn(T + 1, B, device=dev, generator=g)
    r = torch.randn(T, B, device=dev, generator=g)
    ga = torch.randn(T, B, device=dev, generator=g)
    return v, r, ga, torch.empty_like(r), torch.empty_like(v), torch.empty_like(r)


def ktimed(seq, n):
    for f in seq:
        assert f() == 0
    torch.cuda.synchronize()
    tot = len(seq) * n
    assert lib.hpc_rll_ktime_begin(tot) == 0
    for _ in range(n):
        for f in seq:
            f()
    ms = (ctypes.c_float * tot)()
    kd = (ctypes.c_int * tot)()
    assert lib.hpc_rll_ktime_end(ms, kd, tot) == tot
    return [statistics.median([ms[j * len(seq) + i] for j in range(2, n)]) * 1e3 for i in range(len(seq))]


for k in range(3):
    v, r, ga, adv, gv, gr = bufs()
    fwd = lambda c: lib.hpc_rll_gae_forward_ex(v.data_ptr(), r.data_ptr(), adv.data_ptr(), coef.data_ptr(), T, B, 0.99, *c, s)  # noqa: E731
    bwd = lambda c: lib.hpc_rll_gae_backward_ex(ga.data_ptr(), gv.data_ptr(), gr.data_ptr(), coef.data_ptr(), T, B, 0.99, *c, s)  # noqa: E731
    for _ in range(100):
        fwd(AUTO); bwd(AUTO)
    ref = None
    for name, cf, cb in (("shipped", AUTO, AUTO), ("explicit (4,4,8,11) / (4,2,4,10)", (4, 4, 8, 11), (4, 2, 4, 10)),
                         ("XCD-contiguous tiles", (4, 4, 8, 43), (4, 2, 4, 42)), ("XCD tiles, plain stores", (4, 4, 8, 41), (4, 2, 4, 40)),
                         ("shipped", AUTO, AUTO)):
        t = ktimed([lambda: fwd(cf), lambda: bwd(cb)], 40)
        torch.cuda.synchronize()
        out = (adv.clone(), gv.clone(), gr.clone())
        if ref is None:
            ref = out
        same = all(torch.equal(a, b) for a, b in zip(ref, out))
        print(f"set {k}: {name:36s} fwd {t[0]:7.1f} us ({BYTES / t[0] / 1e3:5.0f} GB/s)  bwd {t[1]:7.1f} us ({BYTES / t[1] / 1e3:5.0f} GB/s)  identical {same}", flush=True)
