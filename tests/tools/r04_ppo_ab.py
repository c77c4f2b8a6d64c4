#!/usr/bin/env python3
"""PPO forward: one fused launch (tune key 32 = 1) against two categorical launches + the sample launch (0), at the suite's
shape (B = 65536, N = 128) and a few others, interleaved in one process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import bench_suite as S  # noqa: E402
import hpc_rl_utils as U  # noqa: E402

S.QUIET = True
FOLD = int(os.environ.get("FOLD", "1"))       # tune key 21: 2 = the counter-tree fold for the 2048 / 4096-workgroup launches of key 32 = 2 / 3
U.tune_set(21, FOLD)
for B, N in ((65536, 128), (262144, 18), (65536, 256)):
    for key in (0, 1, 2, 3, 0, 1, 2, 3):
        U.tune_set(32, key)
        S.rows.clear()
        S.suite_ppo(B, N)
        r = S.rows[-1]
        print(("fold=%d " % FOLD) + "B=%d N=%d key32=%d fwd %.4f ms frac %.3f  bwd %.4f | graph replay: fwd %.4f ms frac %.3f  bwd %.4f frac %.3f" %
              (B, N, key, r["fwd_ms"], r["fwd_frac"], r["bwd_ms"], r["fwd_kernel_ms"], r["fwd_kernel_frac"], r["bwd_kernel_ms"],
               r["bwd_kernel_frac"]), flush=True)
U.tune_set(32, 1)
U.tune_set(21, 1)
