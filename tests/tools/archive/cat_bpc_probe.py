#!/usr/bin/env python3
"""categorical forward/backward at the C3 shape: resident blocks per CU (tune key 0), interleaved rounds."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi as N  # noqa: E402
dev = torch.device("cuda:0")
for n in (128, 32, 512):
    rows = 256 * 16384 * 128 // n
    x = torch.randn(rows, n, device=dev)
    a = torch.randint(0, n, (rows,), device=dev)
    logp, ent = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    g = torch.empty(rows, n, device=dev)
    c = torch.randn(rows, device=dev)
    one = torch.ones(1, device=dev)

    def t(fn, k=5):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / k * 1e-3
    best = {}
    for rnd in range(3):
        for bpc in (6, 8, 12, 16, 24, 32):
            N.check(N.lib.hpc_rll_tune_set(0, bpc))
            tf = t(lambda: N.call("hpc_rll_categorical_forward", dev, x.data_ptr(), a.data_ptr(), logp.data_ptr(), ent.data_ptr(), rows, n))
            tb = t(lambda: N.call("hpc_rll_categorical_backward", dev, x.data_ptr(), a.data_ptr(), c.data_ptr(), one.data_ptr(),
                                  c.data_ptr(), one.data_ptr(), g.data_ptr(), rows, n))
            best[bpc] = (min(best.get(bpc, (1e9, 1e9))[0], tf), min(best.get(bpc, (1e9, 1e9))[1], tb))
    by = rows * n * 4
    print(f"N={n}: " + "  ".join(f"bpc={k}: {v[0]*1e6:.0f}/{v[1]*1e6:.0f} us" for k, v in best.items()), flush=True)
    del x, g
N.check(N.lib.hpc_rll_tune_set(0, 24))
