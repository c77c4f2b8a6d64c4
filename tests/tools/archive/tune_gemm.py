#!/usr/bin/env python3
"""fp32 MFMA GEMM: BK=16 vs BK=32 at the LSTM C4 shapes (NN, NT, TN) + correctness vs torch fp64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cabi
import hpc_torch_utils_network as U
dev = torch.device("cuda:0")
def t(fn, n=5):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / n * 1e-3
shapes = [("NN rec", 4096, 4096, 1024, "nn"), ("NT dh", 4096, 1024, 4096, "nt"), ("NN xw", 65536, 4096, 1024, "nn"),
          ("TN dW", 1024, 4096, 65536, "tn"), ("NT dx", 65536, 1024, 4096, "nt"), ("NN sq", 4096, 4096, 4096, "nn")]
for bk in (32, 16):
    assert cabi.lib.hpc_rll_tune_set(1, bk) == 0
    for name, M, N, K, lay in shapes:
        a = torch.randn(M, K, device=dev); b = torch.randn(K, N, device=dev)
        A = a if lay != "tn" else a.t().contiguous().t()
        Bm = b if lay != "nt" else b.t().contiguous().t()
        c = torch.empty(M, N, device=dev)
        dt = t(lambda: U.gemm_f32(A, Bm, out=c))
        ref = (a[:256].double() @ b.double()[:, :256])
        err = (c[:256, :256].double() - ref).abs().max().item() / ref.abs().max().item()
        print(f"BK={bk} {name:7s} M={M} N={N} K={K}: {dt*1e3:8.3f} ms {2.0*M*N*K/dt/1e12:6.1f} TF  relerr {err:.1e}", flush=True)
        del a, b, c, A, Bm
