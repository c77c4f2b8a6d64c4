#!/usr/bin/env python3
"""the mapping BASELINE.json's north_star names -- one trajectory per wavefront, (T,B) tile staged
through LDS, 64-lane shuffle scan along time (gae_fwd_wpt_kernel, flags bit 4) -- A/B'd in-process against the shipped
lane-per-column forward at T=1024, B in {64, 1024, 4096, 65536} (and T=256, B=16384).  Forward and backward alternate;
per-kernel time = the dispatch's own begin/end (hpc_rll_ktime_*), median.  Writes gpurun_out/r03_gae_wpt_probe.txt"""
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
s = torch.cuda.current_stream().cuda_stream
lines = ["GAE forward: shipped lane-per-column mapping vs one-trajectory-per-wavefront (LDS-staged tile + 64-lane shuffle scan)",
         "T B | shipped_us | wave_per_trajectory_us (nontemporal loads / plain loads) | max rel diff | shipped config"]
for T, B in ((1024, 64), (1024, 1024), (1024, 4096), (1024, 65536), (256, 16384), (1000, 100)):
    g = torch.Generator(device=dev).manual_seed(B)
    v = torch.randn(T + 1, B, device=dev, generator=g)
    r = torch.randn(T, B, device=dev, generator=g)
    ga = torch.randn(T, B, device=dev, generator=g)
    adv, adv2, gv, gr = torch.empty_like(r), torch.empty_like(r), torch.empty_like(v), torch.empty_like(r)
    coef = torch.empty(T, device=dev)
    assert lib.hpc_rll_gae_coef(coef.data_ptr(), T, 0.99, 0.97, s) == 0

    def fwd(flags, out):
        return lib.hpc_rll_gae_forward_ex(v.data_ptr(), r.data_ptr(), out.data_ptr(), coef.data_ptr(), T, B, 0.99, 0, 0, 0, flags, s)

    def bwd():
        return lib.hpc_rll_gae_backward(ga.data_ptr(), gv.data_ptr(), gr.data_ptr(), coef.data_ptr(), T, B, 0.99, s)

    def timed(flags, out, n=60):
        for _ in range(5):
            assert fwd(flags, out) == 0 and bwd() == 0
        torch.cuda.synchronize()
        assert lib.hpc_rll_ktime_begin(2 * n) == 0
        for _ in range(n):
            fwd(flags, out); bwd()
        ms = (ctypes.c_float * (2 * n))()
        kd = (ctypes.c_int * (2 * n))()
        assert lib.hpc_rll_ktime_end(ms, kd, 2 * n) == 2 * n
        return statistics.median(ms[2 * i] for i in range(5, n)) * 1e3

    t_ship = timed(-1, adv)
    c6 = (ctypes.c_int * 6)()
    lib.hpc_rll_gae_last_config(0, c6)
    t_wpt_nt = timed(17, adv2)
    t_wpt = timed(16, adv2)
    torch.cuda.synchronize()
    diff = ((adv - adv2).abs() / adv.abs().clamp(min=1.0)).max().item()
    assert diff < 1e-5, diff
    lines.append(f"{T} {B} | {t_ship:8.1f} | {t_wpt_nt:8.1f} / {t_wpt:8.1f} | {diff:.2e} | {list(c6)}")
txt = "\n".join(lines)
print(txt)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "r03_gae_wpt_probe.txt"), "w").write(txt + "\n")
