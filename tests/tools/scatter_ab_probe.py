#!/usr/bin/env python3
"""In-process A/B of two builds of pad_scatter.hip (shipped library vs ALT side library linked -Wl,-Bsymbolic):
ScatterConnection backward at the C5 shape, interleaved rounds."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi as N  # noqa: E402
alt = ctypes.CDLL(os.environ.get("ALT", os.path.join(ROOT, "tests", "tools", "micro", "libsc_old.so")))
name = "hpc_rll_scatter_connection_backward"
getattr(alt, name).argtypes = N.SIGNATURES[name][1]
getattr(alt, name).restype = ctypes.c_int
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B, M, C, H, W = 4096, 256, 64, 64, 64
go = torch.randn(B, C, H, W, device=dev)
loc = torch.stack([torch.randint(0, H, (B, M), device=dev), torch.randint(0, W, (B, M), device=dev)], -1)
gx = torch.empty(B, M, C, device=dev)


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
for rnd in range(5):
    for tag, L in (("new", N.lib), ("old", alt)):
        v = t(lambda: getattr(L, name)(go.data_ptr(), loc.data_ptr(), gx.data_ptr(), B, M, C, H, W, st))
        res[tag] = min(res.get(tag, 1e9), v)
print("scatter backward C5: " + "  ".join(f"{k}: {v:.0f} us" for k, v in res.items()))
