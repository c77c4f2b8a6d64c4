#!/usr/bin/env python3
"""In-process A/B of two builds of categorical.hip: the shipped library vs a side library
(ALT=path, default tests/tools/micro/libcat_old.so built from another revision of the file), interleaved rounds."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi as N  # noqa: E402
alt = ctypes.CDLL(os.environ.get("ALT", os.path.join(ROOT, "tests", "tools", "micro", "libcat_old.so")))
for name in ("hpc_rll_categorical_forward", "hpc_rll_categorical_backward"):
    getattr(alt, name).argtypes = N.SIGNATURES[name][1]
    getattr(alt, name).restype = ctypes.c_int
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
for n in [int(v) for v in os.environ.get("NS", "128,32,512,18,1000").split(",")]:
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
        return e0.elapsed_time(e1) / k * 1e3
    res = {}
    for rnd in range(4):
        for tag, L in (("new", N.lib), ("old", alt)):
            fe = t(lambda: L.hpc_rll_categorical_forward(x.data_ptr(), a.data_ptr(), logp.data_ptr(), ent.data_ptr(), rows, n, st))
            fn = t(lambda: L.hpc_rll_categorical_forward(x.data_ptr(), a.data_ptr(), logp.data_ptr(), None, rows, n, st))
            b = t(lambda: L.hpc_rll_categorical_backward(x.data_ptr(), a.data_ptr(), c.data_ptr(), one.data_ptr(), c.data_ptr(),
                                                         one.data_ptr(), g.data_ptr(), rows, n, st))
            r = res.setdefault(tag, [1e9, 1e9, 1e9])
            res[tag] = [min(r[0], fe), min(r[1], fn), min(r[2], b)]
    print(f"N={n:5d} rows={rows}: " + "   ".join(f"{k}: fwd+ent {v[0]:.0f}  fwd {v[1]:.0f}  bwd {v[2]:.0f} us" for k, v in res.items()), flush=True)
    del x, g
