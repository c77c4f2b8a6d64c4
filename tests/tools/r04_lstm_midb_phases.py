#!/usr/bin/env python3
"""HPC_RLL_LSTM_PROFILE=1: per-phase times of the mid-batch backward kernel (one forward + backward per shape)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402

for B, H in ((64, 1024), (64, 384), (16, 512), (128, 512), (32, 1024)):
    m = LSTM(64, B, H, H, 1).cuda()
    x = torch.randn(64, B, H, device="cuda", requires_grad=True)
    y, _ = m(x, None)
    print(f"B={B} H={H}", file=sys.stderr, flush=True)
    y.sum().backward()
    torch.cuda.synchronize()
