#!/usr/bin/env python3
"""Workload for a PMC pass over the large-batch C51 / QR-DQN forwards (B = 262144, N = 64, 51 atoms / tau = 32): three
launches each through the C ABI, with the samples-per-wave kernels (default) and with tune key 24 = 1 (the wave- / group-
per-sample kernels).  Run under rocprofv3 --pmc by tests/tools/r03_td_pmc.sh."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi as N  # noqa: E402

lib = N.lib
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
P = lambda t: t.data_ptr()  # noqa: E731
B, Nq, nstep, n_atom, tau = 1 << 18, 64, 5, 51, 32
reward = torch.randn(nstep, B, device=dev, generator=g)
done = (torch.rand(B, device=dev, generator=g) < 0.1).float()
weight = torch.rand(B, device=dev, generator=g)
a = torch.randint(0, Nq, (B,), device=dev, generator=g)
na = torch.randint(0, Nq, (B,), device=dev, generator=g)
loss, td = torch.empty(1, device=dev), torch.empty(B, device=dev)
part = torch.empty(int(lib.hpc_rll_partials_floats(B)), device=dev)
d = torch.softmax(torch.randn(B, Nq, n_atom, device=dev, generator=g), -1)
nd = torch.softmax(torch.randn(B, Nq, n_atom, device=dev, generator=g), -1)
buf = torch.empty(B, n_atom, device=dev)
for key in (0, 1):
    lib.hpc_rll_tune_set(24, key)
    for _ in range(3):
        assert lib.hpc_rll_dist_nstep_td_forward(P(d), P(nd), P(a), P(na), P(reward), P(done), P(weight), P(loss), P(td), P(buf),
                                                 P(part), nstep, B, Nq, n_atom, 0.99, -10.0, 10.0, 1.0 / B, s) == 0
    torch.cuda.synchronize()
del d, nd
q = torch.randn(B, Nq, tau, device=dev, generator=g)
nq = torch.randn(B, Nq, tau, device=dev, generator=g)
buf = torch.empty(B, tau, device=dev)
for key in (0, 1):
    lib.hpc_rll_tune_set(24, key)
    for _ in range(3):
        assert lib.hpc_rll_qrdqn_nstep_td_forward(P(q), P(nq), P(a), P(na), P(reward), P(done), P(weight), None, P(loss), P(td),
                                                  P(buf), P(part), tau, nstep, B, Nq, 0.99, 1.0, 1.0 / B, s) == 0
    torch.cuda.synchronize()
lib.hpc_rll_tune_set(24, 0)
print("td pmc probe done")
