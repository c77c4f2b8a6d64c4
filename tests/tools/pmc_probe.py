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
# the launch configuration these counters belong to (bench.py quotes the traffic only for the SAME configuration)
import ctypes  # noqa: E402
import json  # noqa: E402
lib = ctypes.CDLL(os.path.join(ROOT, "di-hpc_amd", "hpc_rll", "_lib", "libhpc_rll_hip.so"))
cfg = {}
for d, name in ((0, "gae_fwd_kernel"), (1, "gae_bwd_kernel")):
    c6 = (ctypes.c_int * 6)()
    lib.hpc_rll_gae_last_config(d, c6)
    cfg[name] = {"cols_per_lane": c6[0], "steps_per_chunk": c6[1], "waves_per_workgroup": c6[2], "nontemporal": c6[3],
                 "half_wave_tiles": bool(c6[4]), "pipelined": bool(c6[5])}
if os.environ.get("PROBE_CONFIG_OUT"):
    json.dump(cfg, open(os.environ["PROBE_CONFIG_OUT"], "w"))
print("probe done: copy bytes R=W=%d ; gae algorithmic bytes per launch=%d" % (T * B * 4, 12 * T * B + 4 * B))
