#!/usr/bin/env python3
"""Workload for the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs): a calibration copy of a
known byte count, then a few launches of each GAE kernel at the headline shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_rl_utils as U  # noqa: E402

T, B = int(os.environ.get("PROBE_T", 1024)), int(os.environ.get("PROBE_B", 65536))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
v = torch.randn(T + 1, B, device=dev, generator=g)
r = torch.randn(T, B, device=dev, generator=g)
ga = torch.randn(T, B, device=dev, generator=g)
adv, gv, gr = torch.empty_like(r), torch.empty_like(v), torch.empty_like(r)
# calibration: float4 copy of exactly T*B floats read + T*B floats written (torch's vectorised copy kernel)
dst = torch.empty_like(r)
for _ in range(3):
    dst.copy_(r)
for _ in range(3):
    U.GaeForward([v, r], [adv], 0.99, 0.97)
for _ in range(3):
    U.GaeBackward([ga], [gv, gr], 0.99, 0.97)
torch.cuda.synchronize()
print("probe done: copy bytes R=W=%d ; gae algorithmic bytes per launch=%d" % (T * B * 4, 12 * T * B + 4 * B))
