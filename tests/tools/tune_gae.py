#!/usr/bin/env python3
"""GPU-side tuning sweep for the GAE kernels: times every (vec, lc, nw) instantiation through the C ABI's
expert entry points at the headline shape and writes a table to gpurun_out/tune_gae.{txt,json}."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import hpc_rl_utils as U  # noqa: E402
import cabi  # noqa: E402

T = int(os.environ.get("TUNE_T", 1024))
B = int(os.environ.get("TUNE_B", 65536))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
v = torch.randn(T + 1, B, device=dev, generator=g)
r = torch.randn(T, B, device=dev, generator=g)
ga = torch.randn(T, B, device=dev, generator=g)
adv, gv, gr = torch.empty_like(r), torch.empty_like(v), torch.empty_like(r)
coef = U.gae_coef(T, 0.99, 0.97, dev)
lib, s = cabi.lib, cabi.stream_ptr(dev)
BYTES = 12 * T * B + 4 * B


def timed(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


rows = []
# reference points: device copy bandwidth on this box
src = torch.empty(BYTES // 8, device=dev); dst = torch.empty_like(src)
t = timed(lambda: dst.copy_(src))
rows.append(("copy_", 0, 0, 0, t, t))
import statistics
acc = {}
ROUNDS = int(os.environ.get("ROUNDS", 3))
for rnd in range(ROUNDS):
    for vec in (1, 2, 4):
        for lc in (2, 4, 8, 16):
            for nw in (1, 2, 4, 8, 16):
                for fl in (0, 1, 2, 3):
                    f = lambda: lib.hpc_rll_gae_forward_ex(v.data_ptr(), r.data_ptr(), adv.data_ptr(), coef.data_ptr(), T, B, 0.99, vec, lc, nw, fl, s)
                    b = lambda: lib.hpc_rll_gae_backward_ex(ga.data_ptr(), gv.data_ptr(), gr.data_ptr(), coef.data_ptr(), T, B, 0.99, vec, lc, nw, fl, s)
                    if f() != 0 or b() != 0:
                        continue
                    acc.setdefault((vec, lc, nw, fl), []).append((timed(f, 5), timed(b, 5)))
for (vec, lc, nw, fl), ts in acc.items():
    rows.append((f"nt{fl}", vec, lc, nw, statistics.median(t[0] for t in ts), statistics.median(t[1] for t in ts)))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
lines = [f"T={T} B={B} algorithmic bytes/launch={BYTES}", "kind vec lc nw   fwd_us  fwd_GB/s   bwd_us  bwd_GB/s"]
lines.append("--- sorted by forward time")
for k, vec, lc, nw, tf, tb in sorted(rows, key=lambda x: x[4])[:25]:
    lines.append(f"{k:5s} {vec:3d} {lc:2d} {nw:2d} {tf*1e6:8.1f} {BYTES/tf/1e9:9.0f} {tb*1e6:8.1f} {BYTES/tb/1e9:9.0f}")
lines.append("--- sorted by backward time")
for k, vec, lc, nw, tf, tb in sorted(rows, key=lambda x: x[5])[:25]:
    lines.append(f"{k:5s} {vec:3d} {lc:2d} {nw:2d} {tf*1e6:8.1f} {BYTES/tf/1e9:9.0f} {tb*1e6:8.1f} {BYTES/tb/1e9:9.0f}")
lines.append("--- all, sorted by sum")
for k, vec, lc, nw, tf, tb in sorted(rows, key=lambda x: x[4] + x[5]):
    lines.append(f"{k:5s} {vec:3d} {lc:2d} {nw:2d} {tf*1e6:8.1f} {BYTES/tf/1e9:9.0f} {tb*1e6:8.1f} {BYTES/tb/1e9:9.0f}")
txt = "\n".join(lines)
print(txt)
tag = f"_{T}x{B}"
open(os.path.join(ROOT, "gpurun_out", f"tune_gae{tag}.txt"), "w").write(txt + "\n")
json.dump([dict(kind=k, vec=a, lc=b_, nw=c, fwd_s=tf, bwd_s=tb) for k, a, b_, c, tf, tb in rows],
          open(os.path.join(ROOT, "gpurun_out", f"tune_gae{tag}.json"), "w"))
