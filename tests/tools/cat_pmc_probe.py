#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc on the categorical forward/backward kernels at the C3 shape."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi as N  # noqa: E402
dev = torch.device("cuda:0")
rows, n = 256 * 16384, 128
x = torch.randn(rows, n, device=dev)
a = torch.randint(0, n, (rows,), device=dev)
logp, ent = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
g = torch.empty(rows, n, device=dev)
c = torch.randn(rows, device=dev)
one = torch.ones(1, device=dev)
for _ in range(3):
    N.call("hpc_rll_categorical_forward", dev, x.data_ptr(), a.data_ptr(), logp.data_ptr(), ent.data_ptr(), rows, n)
for _ in range(3):
    N.call("hpc_rll_categorical_backward", dev, x.data_ptr(), a.data_ptr(), c.data_ptr(), one.data_ptr(), c.data_ptr(),
           one.data_ptr(), g.data_ptr(), rows, n)
torch.cuda.synchronize()
print("done")
