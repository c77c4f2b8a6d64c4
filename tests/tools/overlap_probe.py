#!/usr/bin/env python3
"""Would overlapping the C4 x-branch GEMM (independent of the recurrence) with the recurrent loop pay?  Sequential vs
two streams (recurrence on a high-priority stream, x-branch chunks on a low-priority one, an event per chunk)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_torch_utils_network as U  # noqa: E402
dev = torch.device("cuda:0")
S, B, I, H = 128, 4096, 1024, 1024
G = 4 * H
x = torch.randn(S * B, I, device=dev)
wx = torch.randn(I, G, device=dev)
wh = torch.randn(H, G, device=dev)
xw = torch.empty(S * B, G, device=dev)
h = torch.randn(B, H, device=dev)
hw = torch.empty(B, G, device=dev)
gates = torch.empty(B, G, device=dev)
CH = 8                                   # x-branch chunks


def recur_step(s):
    U.gemm_f32(h, wh, out=hw)
    torch.add(hw, xw[s * B:(s + 1) * B], out=gates)     # stand-in for the cell kernel (memory bound, ~same bytes)


def sequential():
    U.gemm_f32(x, wx, out=xw)
    for s in range(S):
        recur_step(s)


hi = torch.cuda.Stream(priority=-1)
lo = torch.cuda.Stream(priority=0)


def overlapped():
    cur = torch.cuda.current_stream()
    hi.wait_stream(cur)
    lo.wait_stream(cur)
    evs = []
    rows = S * B // CH
    with torch.cuda.stream(lo):
        for c in range(CH):
            U.gemm_f32(x[c * rows:(c + 1) * rows], wx, out=xw[c * rows:(c + 1) * rows])
            e = torch.cuda.Event()
            e.record(lo)
            evs.append(e)
    with torch.cuda.stream(hi):
        for s in range(S):
            if s % (S // CH) == 0:
                hi.wait_event(evs[s // (S // CH)])
            recur_step(s)
    cur.wait_stream(hi)
    cur.wait_stream(lo)


def t(fn, n=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


for rnd in range(2):
    print(f"sequential {t(sequential):.1f} ms   overlapped {t(overlapped):.1f} ms", flush=True)
