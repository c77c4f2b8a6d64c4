#!/usr/bin/env python3
"""Workload for the VALU-instruction count of the instruction-bound forwards at the shapes bench.py's suite runs them
(tests/tools/bench_suite.py: suite_td): C51 and QR-DQN at B = 262144, IQN at B = 65536; N = 64, 51 atoms / tau = 32, nstep 5.
Three launches each through the C ABI, default kernels.  Run under rocprofv3 --pmc by tests/tools/r04_td_valu.sh."""
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
for _ in range(3):
    assert lib.hpc_rll_dist_nstep_td_forward(P(d), P(nd), P(a), P(na), P(reward), P(done), P(weight), P(loss), P(td), P(buf),
                                             P(part), nstep, B, Nq, n_atom, 0.99, -10.0, 10.0, 1.0 / B, s) == 0
torch.cuda.synchronize()
del d, nd
q = torch.randn(B, Nq, tau, device=dev, generator=g)
nq = torch.randn(B, Nq, tau, device=dev, generator=g)
buf = torch.empty(B, tau, device=dev)
for _ in range(3):
    assert lib.hpc_rll_qrdqn_nstep_td_forward(P(q), P(nq), P(a), P(na), P(reward), P(done), P(weight), None, P(loss), P(td),
                                              P(buf), P(part), tau, nstep, B, Nq, 0.99, 1.0, 1.0 / B, s) == 0
torch.cuda.synchronize()
del q, nq
Bi = B // 4
qi = torch.randn(tau, Bi, Nq, device=dev, generator=g)
nqi = torch.randn(tau, Bi, Nq, device=dev, generator=g)
rq = torch.rand(tau, Bi, device=dev, generator=g)
ri, di, wi, ai, nai = reward[:, :Bi].contiguous(), done[:Bi].contiguous(), weight[:Bi].contiguous(), a[:Bi].contiguous(), na[:Bi].contiguous()
bufi = torch.empty(Bi, tau, device=dev)
for _ in range(3):
    assert lib.hpc_rll_iqn_nstep_td_forward(P(qi), P(nqi), P(ai), P(nai), P(ri), P(di), P(rq), P(wi), None, P(loss), P(td), P(bufi),
                                            P(part), tau, tau, nstep, Bi, Nq, 0.99, 1.0, 1.0 / Bi, s) == 0
torch.cuda.synchronize()
print("td valu probe done")
