import os, sys, statistics, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cabi as C
dev = torch.device("cuda:0")
def t(fn, k=5, rounds=3):
    fn(); ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / k * 1e3)
    return statistics.median(ts)
T, B, N = 256, 16384, 128
rows = T * B
a = torch.zeros(rows, dtype=torch.int64, device=dev)
lp, en = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
bufs = []
for k in range(24):
    bufs.append(torch.randn(rows, N, device=dev))
for k, x in enumerate(bufs):
    tc = t(lambda: C.call("hpc_rll_categorical_forward", dev, x.data_ptr(), a.data_ptr(), lp.data_ptr(), en.data_ptr(), rows, N))
    p = x.data_ptr()
    print(f"buf {k:2d} va {p:#x}  va>>31 = {p >> 31:#x} (mod 4GB: {(p >> 31) & 1})  {tc:.0f} us", flush=True)
# one 8 GB buffer, 2 GB windows at 512 MB offsets
big = torch.randn(4 * rows + rows // 2 * 0, N, device=dev)
for off in range(0, 3 * rows + 1, rows // 4):
    x = big[off:off + rows]
    if x.shape[0] < rows: break
    tc = t(lambda: C.call("hpc_rll_categorical_forward", dev, x.data_ptr(), a.data_ptr(), lp.data_ptr(), en.data_ptr(), rows, N))
    print(f"big+{off * N * 4 / 2**30:.2f} GiB  va {x.data_ptr():#x}  {tc:.0f} us", flush=True)
