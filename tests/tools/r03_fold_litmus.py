#!/usr/bin/env python3
"""a recorded litmus for the folded loss finalisation (csrc/colscan.hpp: publish_sums / ScanFold).

The last workgroup of a scalar-loss launch adds the other workgroups' partial sums.  The hand-over is a relaxed
agent-scope atomic store of the partial + s_waitcnt vmcnt(0) + barrier + a relaxed agent-scope ticket (no release
fence: an agent-scope release would write the XCD's whole dirty L2 back) -- outside the HSA memory
model, so it is validated empirically: >= 10^5 folded launches spread over 8 streams (every XCD populated: 128+
workgroups per launch), three kernel families (column scan: TD-lambda; per-sample: QR-DQN; categorical + scan:
V-trace), each result compared BIT FOR BIT with the separate-finalize launch (tune key 21 = 0) on the same data.
Writes gpurun_out/r03_fold_litmus.json."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_rl_utils as U  # noqa: E402
from hpc_rll.rl_utils.td import QRDQNNStepTDError, TDLambda  # noqa: E402
from hpc_rll.rl_utils.vtrace import VTrace  # noqa: E402

dev = torch.device("cuda:0")
NSTREAM = int(os.environ.get("LITMUS_STREAMS", 8))
PER = int(os.environ.get("LITMUS_PER_STREAM", 4200))     # x 3 ops x 8 streams = 100,800 folded launches
T, B, N = 24, 8192, 4
g = torch.Generator(device=dev).manual_seed(5)
data = [(torch.randn(T + 1, B, device=dev, generator=g), torch.randn(T, B, device=dev, generator=g),
         torch.randn(T, B, N, device=dev, generator=g), torch.randn(T, B, N, device=dev, generator=g),
         torch.randint(0, N, (T, B), device=dev, generator=g)) for _ in range(NSTREAM)]
td, vt, qr = TDLambda(T, B), VTrace(T, B, N), QRDQNNStepTDError(N, 3, B, T)
done = torch.zeros(B, device=dev)
qs = [d[2].transpose(0, 1).contiguous() for d in data]


def losses(i):
    v, r, to, bo, a = data[i]
    q = qs[i]
    return [td(v, r), *vt(to, bo, a, v, r), qr(q, q.flip(0), a[0] % T, a[1] % T, r[:3].contiguous(), done, 0.9)[0]]


with torch.no_grad():
    U.tune_set(21, 0)
    want = [torch.cat([x.reshape(1) for x in losses(i)]) for i in range(NSTREAM)]
    U.tune_set(21, 1)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(NSTREAM)]
    got = [torch.empty(PER, 5, device=dev) for _ in range(NSTREAM)]
    t0 = time.time()
    for k in range(PER):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                ls = losses(i)
                torch.stack([x.reshape(()) for x in ls], out=got[i][k])
    torch.cuda.synchronize()
    dt = time.time() - t0
bad = 0
for i in range(NSTREAM):
    bad += int((got[i] != want[i][None, :]).any(dim=1).sum().item())
res = {"folded_launches": 3 * PER * NSTREAM, "streams": NSTREAM, "launches_per_stream": 3 * PER,
       # tau=4: 8-lane groups, 32 samples per 256-thread workgroup (dist_ops.hip); all <= kFoldMaxGrid
       "workgroups_per_launch": {"td_lambda": "128+ (B=8192 columns)", "vtrace_scan": "128+", "qrdqn": B // 32},
       "mismatching_results": bad, "seconds": dt,
       "compared_with": "separate finalize launch (hpc_rll_tune_set(21, 0)), bit for bit, same partials",
       "device": torch.cuda.get_device_name(0)}
print(json.dumps(res))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r03_fold_litmus.json"), "w"), indent=1)
assert bad == 0
