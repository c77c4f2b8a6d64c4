#!/usr/bin/env python3
"""Round 6: TD-lambda forward at C3 (T=256, B=16384) by scan configuration (HPC_RLL_TD_CFG = v,lc,nw,sub), kernel time by hipGraph
replay (10 launches per graph), loss and unit gradient compared with the shipped configuration."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench_suite as S
import hpc_rl_utils as U
dev = torch.device("cuda:0")
for (T, B) in ((256, 16384), (256, 8192), (128, 16384), (512, 16384), (256, 32768)):
    g = torch.Generator(device=dev).manual_seed(0)
    value = torch.randn(T + 1, B, device=dev, generator=g); reward = torch.randn(T, B, device=dev, generator=g)
    weight = torch.rand(T, B, device=dev, generator=g)
    loss, gb = torch.empty(1, device=dev), torch.empty(T, B, device=dev)
    fn = lambda: U.TdLambdaForward([value, reward, weight], [loss, gb], 0.9, 0.8)
    ref = None
    out = []
    for cfg in (None, "1,16,16,1", "1,8,16,2", "1,8,16,4", "1,8,16,1", "1,8,8,1", "2,8,16,1", "2,8,8,1", "1,8,4,1"):
        if cfg is None: os.environ.pop("HPC_RLL_TD_CFG", None)
        else: os.environ["HPC_RLL_TD_CFG"] = cfg
        fn(); torch.cuda.synchronize()
        if ref is None: ref = (loss.clone(), gb.clone())
        else:
            assert float((gb - ref[1]).abs().max()) <= 2e-6 * float(ref[1].abs().max()), (cfg, float((gb - ref[1]).abs().max()))
            assert abs(loss.item() - ref[0].item()) <= 2e-6 * abs(ref[0].item()), (cfg, loss.item(), ref[0].item())
        t = min(S.timed_graph(fn, n=10) for _ in range(3))
        out.append(f"{cfg or 'shipped'}: {t*1e6:.1f}")
    os.environ.pop("HPC_RLL_TD_CFG", None)
    print(f"T={T} B={B} ({16*T*B/1e6:.0f} MB): " + "  ".join(out) + " us", flush=True)
