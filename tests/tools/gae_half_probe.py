#!/usr/bin/env python3
"""GAE at narrow batches: half-wave tiles (flags bit 2) vs the 64-column tiling, forward/backward alternating;
results must be bit-identical (same per-column arithmetic order inside a chunk, same chunk size)."""
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
for (T, B) in [(1024, 8192), (1024, 4096), (1024, 64), (1024, 1000), (4096, 2048), (600, 777)]:
    g = torch.Generator(device=dev).manual_seed(0)
    v = torch.randn(T + 1, B, device=dev, generator=g)
    r = torch.randn(T, B, device=dev, generator=g)
    ga = torch.randn(T, B, device=dev, generator=g)
    coef = torch.empty(T, device=dev)
    assert lib.hpc_rll_gae_coef(coef.data_ptr(), T, 0.99, 0.97, s) == 0
    res = {}
    for name, cfg in (("tile64", (1, 16, 16, 2)), ("half32", (1, 16, 16, 2 | 4)), ("auto", (0, 0, 0, -1))):
        adv, gv, gr = torch.empty_like(r), torch.empty_like(v), torch.empty_like(r)
        fwd = lambda: lib.hpc_rll_gae_forward_ex(v.data_ptr(), r.data_ptr(), adv.data_ptr(), coef.data_ptr(), T, B, 0.99, *cfg, s)
        bwd = lambda: lib.hpc_rll_gae_backward_ex(ga.data_ptr(), gv.data_ptr(), gr.data_ptr(), coef.data_ptr(), T, B, 0.99, *cfg, s)
        assert fwd() == 0 and bwd() == 0
        n = 20
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n + 1)]
        ev[0].record()
        for i in range(n):
            fwd(); ev[2 * i + 1].record(); bwd(); ev[2 * i + 2].record()
        ev[-1].synchronize()
        tf = statistics.median(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(1, n)) * 1e3
        tb = statistics.median(ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(1, n)) * 1e3
        res[name] = (adv, gv, gr)
        print(f"T={T} B={B} {name:7s}: fwd {tf:6.1f} us  bwd {tb:6.1f} us", flush=True)
    for a, b in zip(res["tile64"], res["half32"]):
        assert torch.equal(a, b), "half-wave tiles changed the result"
print("half-wave tiles: bit-identical")
