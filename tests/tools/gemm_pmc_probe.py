#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes on the fp32 MFMA GEMM at the C4 LSTM shapes (a few launches each)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_torch_utils_network as U  # noqa: E402
dev = torch.device("cuda:0")
shapes = [("NN rec", 4096, 4096, 1024, "nn"), ("NT dh", 4096, 1024, 4096, "nt"), ("TN dW", 1024, 4096, 65536, "tn"),
          # round 3: shapes that take the LDS-DMA kernels without split-K (the LSTM's recurrent product as it now runs, and a
          # TN product of 256 workgroups)
          ("NT rec (LDS-DMA 256x128)", 4096, 4096, 1024, "nt"), ("TN 4096^3 (LDS-DMA k-major)", 4096, 4096, 4096, "tn")]
for name, M, N, K, lay in shapes:
    a = torch.randn(M, K, device=dev)
    b = torch.randn(K, N, device=dev)
    A = a if lay != "tn" else a.t().contiguous().t()
    Bm = b if lay != "nt" else b.t().contiguous().t()
    c = torch.empty(M, N, device=dev)
    for _ in range(3):
        U.gemm_f32(A, Bm, out=c)
    torch.cuda.synchronize()
    del a, b, c, A, Bm
print("done")
