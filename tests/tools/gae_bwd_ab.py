#!/usr/bin/env python3
"""Headline shape: backward configurations interleaved in one process (forward = auto), alternating fwd/bwd."""
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
T, B = 1024, 65536
g = torch.Generator(device=dev).manual_seed(0)
v = torch.randn(T + 1, B, device=dev, generator=g)
r = torch.randn(T, B, device=dev, generator=g)
ga = torch.randn(T, B, device=dev, generator=g)
adv, gv, gr = torch.empty_like(r), torch.empty_like(v), torch.empty_like(r)
coef = torch.empty(T, device=dev)
assert lib.hpc_rll_gae_coef(coef.data_ptr(), T, 0.99, 0.97, s) == 0
fwd = lambda c: lib.hpc_rll_gae_forward_ex(v.data_ptr(), r.data_ptr(), adv.data_ptr(), coef.data_ptr(), T, B, 0.99, *c, s)
bwd = lambda c: lib.hpc_rll_gae_backward_ex(ga.data_ptr(), gv.data_ptr(), gr.data_ptr(), coef.data_ptr(), T, B, 0.99, *c, s)
AUTO = (0, 0, 0, -1)
cands_b = [AUTO, (4, 4, 4, 2), (2, 2, 4, 2), (4, 2, 4, 2), (2, 4, 4, 2), (4, 2, 8, 2), (2, 2, 8, 2), (2, 4, 2, 2), (4, 4, 2, 2)]
cands_f = [AUTO, (2, 8, 2, 3), (2, 8, 4, 3), (2, 4, 4, 3), (4, 8, 2, 3), (2, 16, 2, 3), (4, 4, 4, 3), (2, 8, 2, 2)]


def pair(cf, cb, n=12):
    assert fwd(cf) == 0 and bwd(cb) == 0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n + 1)]
    ev[0].record()
    for i in range(n):
        fwd(cf); ev[2 * i + 1].record(); bwd(cb); ev[2 * i + 2].record()
    ev[-1].synchronize()
    return (statistics.median(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(1, n)) * 1e3,
            statistics.median(ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(1, n)) * 1e3)


res_b = {c: [] for c in cands_b}
res_f = {c: [] for c in cands_f}
for rnd in range(4):
    for c in cands_b:
        res_b[c].append(pair(AUTO, c))
    for c in cands_f:
        res_f[c].append(pair(c, AUTO))
print("backward candidates (fwd auto): cfg -> median bwd us [median fwd us alongside]")
for c, xs in sorted(res_b.items(), key=lambda kv: statistics.median(x[1] for x in kv[1])):
    print(f"   {str(c):18s} bwd {statistics.median(x[1] for x in xs):6.1f}  (fwd {statistics.median(x[0] for x in xs):6.1f})  sum {statistics.median(x[0] + x[1] for x in xs):6.1f}")
print("forward candidates (bwd auto):")
for c, xs in sorted(res_f.items(), key=lambda kv: statistics.median(x[0] for x in kv[1])):
    print(f"   {str(c):18s} fwd {statistics.median(x[0] for x in xs):6.1f}  (bwd {statistics.median(x[1] for x in xs):6.1f})  sum {statistics.median(x[0] + x[1] for x in xs):6.1f}")
