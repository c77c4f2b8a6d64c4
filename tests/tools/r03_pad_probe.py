#!/usr/bin/env python3
"""Where the packed Pad1D call spends its time at n = 2^20 rows: table build, pad kernel, allocations."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cabi  # noqa: E402
import hpc_rl_utils as U  # noqa: E402
from hpc_rll.rl_utils import padding as P  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=10, rounds=5):
    fn()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)


for n, L in ((1 << 20, 127), (1 << 20, 128), (1 << 17, 127)):
    lens = torch.from_numpy(np.random.default_rng(1).integers(32, 128, n)).to(dev)
    flat = torch.randn(int(lens.sum().item()), device=dev)
    t_api = timed(lambda: P.Padding1DPacked(flat, lens, max_len=L))
    new_x, mask = P.Padding1DPacked(flat, lens, max_len=L)
    table = torch.empty(n, 4, dtype=torch.int64, device=dev)
    scratch = torch.empty(int(cabi.lib.hpc_rll_packed_table_scratch_int64(n)), dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    t_tab = timed(lambda: cabi.lib.hpc_rll_packed_table(lens.data_ptr(), n, flat.data_ptr(), 4, table.data_ptr(), scratch.data_ptr(), s))
    t_pad = timed(lambda: cabi.lib.hpc_rll_pad_forward(table.data_ptr(), new_x.data_ptr(), mask.data_ptr(), n, 1, 1, L, 0, s))
    t_lds = timed(lambda: cabi.lib.hpc_rll_pad1d_packed_forward(flat.data_ptr(), table.data_ptr(), new_x.data_ptr(), mask.data_ptr(), n, L, 0, s))
    t_alloc = timed(lambda: (torch.empty(n, L, device=dev), torch.empty(n, L, dtype=torch.int32, device=dev)))
    by = 4 * flat.numel() + 8 * n * L
    t_fill = timed(lambda: (new_x.fill_(0.0), mask.fill_(0)))
    t_un = timed(lambda: P.UnPadding1DPacked(new_x, lens, total=flat.numel()))
    print(f"n={n} L={L}: api {t_api:.1f} us | table {t_tab:.1f} | generic pad kernel {t_pad:.1f} ({by/t_pad/1e3:.0f} GB/s) | packed LDS kernel {t_lds:.1f} ({by/t_lds/1e3:.0f} GB/s) | 2 empty {t_alloc:.1f} | "
          f"2 fills of the outputs {t_fill:.1f} ({8*n*L/t_fill/1e3:.0f} GB/s) | unpad api {t_un:.1f}")
