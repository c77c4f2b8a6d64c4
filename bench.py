#!/usr/bin/env python3
"""Headline benchmark: GAE forward+backward samples/s at T=1024, B=65536 fp32 per GPU (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            # N=1 direct; N>1 under torch.distributed.run

A "step" is one pass of the hot path over one batch of synthetic input: ``adv = GAE(value, reward)``
followed by ``adv.backward(grad_adv)`` through the drop-in ``hpc_rll.rl_utils.gae.GAE`` module (HIP
kernels behind the C ABI).  Inputs are resident in HBM before the timed region.  The batch axis shards
across ranks with NO data-path collective (every trajectory is independent), so per-GPU
work is fixed as N grows: weak scaling; ``value`` = (N * T * B) * K / max-over-ranks wall time.
``--scaling strong`` makes the headline ``value`` the strong reading instead (``--B`` is then the GLOBAL batch, split
over the ranks) and ``--graph`` replays the step from a captured hipGraph -- the launch-latency regime strong scaling
ends up in.

Whatever the headline is, the same JSON line carries ``scaling_detail`` with BOTH readings BASELINE.md section 4 /
SURVEY.md 8d ask for, measured in this run after the headline region: weak (B per GPU) and strong (global B = 65536
split over the N ranks), each eager and as hipGraph replay, with per-rank step times and the RCCL world size seen.
With one rank it also times the per-rank shapes an N = 2, 4, 8 strong-scaling run would hold (B = 32768, 16384, 8192),
so the launch-latency regime is visible without a multi-GPU box.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     -- dominant kernel's algorithmic bytes per launch / its average launch duration measured
                  live with HIP events on the launch stream, vs the 8 TB/s HBM3E peak.
  cpu_baseline -- oracle/gae_ref.c (a port of the reference algorithm, OpenMP) timed on this host, on a
                  bounded sample of the same workload.  A reported baseline, not the target.  Its key
                  `pytorch_restatement` is the same algorithm run the way hpc_rll.origin runs it (oracle/ref_torch.py:
                  a python loop of fp32 torch CPU ops + autograd backward), on a smaller bounded sample.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))

import torch  # noqa: E402

T_DEFAULT, B_DEFAULT = 1024, 65536
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(T, B, gamma, lam, budget_s=12.0):
    """Time the C port of the reference GAE (fwd + adjoint bwd) on the host cores.  oracle/ is used here only
    as the reported CPU baseline."""
    so = os.path.join(ROOT, "oracle", "_build", "libgae_ref.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.gae_ref_forward.argtypes = [fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float]
    lib.gae_ref_backward.argtypes = [fp, fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float]
    lib.gae_ref_num_threads.restype = ctypes.c_int
    Bs = min(B, 16384)  # bounded sample: a quarter of the batch axis (columns are independent)
    g = torch.Generator().manual_seed(0)
    v, r, ga = torch.randn(T + 1, Bs, generator=g), torch.randn(T, Bs, generator=g), torch.randn(T, Bs, generator=g)
    adv, gv, gr, tab = torch.empty(T, Bs), torch.empty(T + 1, Bs), torch.empty(T, Bs), torch.empty(T)
    P = lambda t: ctypes.cast(t.data_ptr(), fp)  # noqa: E731

    def one():
        lib.gae_ref_forward(P(v), P(r), P(adv), T, Bs, gamma, lam)
        lib.gae_ref_backward(P(ga), P(gv), P(gr), P(tab), T, Bs, gamma, lam)

    one()
    reps, t0 = 0, time.perf_counter()
    while True:
        one()
        reps += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or reps >= 200:
            break
    res = {"value": T * Bs * reps / dt, "unit": "samples/s", "cores": int(lib.gae_ref_num_threads()),
           "kind": "port", "sample": f"T={T} B={Bs} (1/{B // Bs} of the batch axis) x {reps} fwd+bwd passes, "
                                     f"oracle/gae_ref.c OpenMP, {dt:.1f}s"}
    # The same algorithm the way hpc_rll.origin runs it (north_star: "next to hpc_rll.origin timed on the same box's host
    # CPU"): the pure-PyTorch restatement oracle/ref_torch.gae (origin/gae.py:28-37, a python loop over T of fp32 tensor
    # ops) forward + autograd backward, on a smaller bounded sample.  Reported beside the (faster) C port, never the target.
    try:
        sys.path.insert(0, ROOT)
        from oracle import ref_torch
        Bt = min(B, 4096)
        vt = torch.randn(T + 1, Bt, generator=g, requires_grad=True)
        rt = torch.randn(T, Bt, generator=g, requires_grad=True)
        gt = torch.randn(T, Bt, generator=g)

        def one_torch():
            vt.grad = rt.grad = None
            ref_torch.gae(vt, rt, gamma, lam).backward(gt)

        one_torch()
        reps_t, t0 = 0, time.perf_counter()
        while True:
            one_torch()
            reps_t += 1
            dt_t = time.perf_counter() - t0
            if dt_t > 6.0 or reps_t >= 50:
                break
        res["pytorch_restatement"] = {"value": T * Bt * reps_t / dt_t, "unit": "samples/s", "cores": torch.get_num_threads(),
                                      "sample": f"T={T} B={Bt} x {reps_t} fwd + autograd bwd passes, oracle/ref_torch.py "
                                                f"(torch {torch.__version__} CPU, fp32), {dt_t:.1f}s"}
    except Exception as e:  # the reported baseline above does not depend on this extra reading
        res["pytorch_restatement"] = {"error": repr(e)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--T", type=int, default=T_DEFAULT)
    ap.add_argument("--B", type=int, default=B_DEFAULT, help="batch per GPU")
    ap.add_argument("--skip-cpu-baseline", action="store_true", help="for profiler runs")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): B per GPU fixed; strong: --B is the GLOBAL batch, split over the ranks")
    ap.add_argument("--no-scaling-detail", action="store_true", help="skip the extra weak/strong legs (profiler runs)")
    ap.add_argument("--graph", action="store_true",
                    help="capture one fwd+bwd step in a hipGraph and replay it (the latency regime: small B per GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hpc_rll product path has no CPU fallback")
    # TEST HOOKS (tests/test_dist.py runs this file with 2 ranks on a ONE-GPU box, where RCCL refuses two ranks per
    # device): HPC_RLL_BENCH_ONE_DEVICE=1 puts every rank on cuda:0, HPC_RLL_BENCH_BACKEND=gloo swaps the backend.  The
    # driver's runs set neither: one rank per GPU over RCCL.
    one_device = os.environ.get("HPC_RLL_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("HPC_RLL_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", 0 if one_device else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or "RANK" in os.environ:   # under torch.distributed.run (also with one rank: same code path)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    import hpc_rl_utils as U
    from hpc_rll.rl_utils.gae import GAE

    T, B, gamma, lam = args.T, args.B, 0.99, 0.97
    global_B = B if args.scaling == "strong" else B * world
    if args.scaling == "strong":
        assert B % world == 0, "strong scaling: the global batch must divide by the number of ranks"
        B //= world

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def make_step(Bk, graph):
        """(step callable, tensors) for one fwd+bwd pass at batch Bk on this rank; graph=True replays a hipGraph."""
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        value = torch.randn(T + 1, Bk, device=dev, generator=g).requires_grad_(True)
        reward = torch.randn(T, Bk, device=dev, generator=g).requires_grad_(True)
        grad_adv = torch.randn(T, Bk, device=dev, generator=g)
        gae = GAE(T, Bk).to(dev)

        def step():
            value.grad = None
            reward.grad = None
            adv = gae(value, reward, gamma, lam)
            adv.backward(grad_adv)

        if not graph:
            return step, (value, reward, grad_adv)
        # same kernels, same order; only the host-side launch path changes (one hipGraphLaunch per step)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        value.grad = None
        reward.grad = None
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            gae(value, reward, gamma, lam).backward(grad_adv)
        return cg.replay, (value, reward, grad_adv, cg)

    def timed(step, steps, warmup):
        """W untimed steps, barrier + synchronize, EXACTLY `steps` timed steps, barrier + synchronize.
        Returns (max over ranks, per-rank list) of the elapsed seconds."""
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        mine = time.perf_counter() - t0
        if dist is None:
            return mine, [mine]
        t = torch.tensor([mine], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [x.item() for x in allt]
        return max(per_rank), per_rank

    step, keep = make_step(B, args.graph)
    value, reward, grad_adv = keep[:3]
    # device clock / allocator pre-roll: a freshly leased GPU idles at its low power state and the first ~10 ms of
    # work run at ramping clocks.  Untimed, not part of W or K (the W warmup steps and the K timed steps follow).
    for _ in range(30):
        step()
    elapsed, per_rank_s = timed(step, args.steps, args.warmup)

    # ---- per-kernel durations, measured IMMEDIATELY after the timed region (same clocks / thermal state: on a power-
    # capped part the step time drifts by ~5 % over the first seconds of streaming): HIP events on the launch stream
    # (torch's current stream) around EVERY launch of a second, instrumented pass over the same K steps in the same
    # alternating fwd/bwd order as the timed region (a kernel repeated back to back would find its inputs in the 256 MiB Infinity Cache and look
    # faster than it is in the real sequence; rocprofv3 --kernel-trace of this command sees the same pattern).
    v_d, r_d = value.detach(), reward.detach()
    adv = torch.empty_like(r_d)
    gv, gr = torch.empty_like(v_d), torch.empty_like(r_d)
    n_ev = max(args.steps, 10)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n_ev + 1)]
    U.GaeForward([v_d, r_d], [adv], gamma, lam)
    U.GaeBackward([grad_adv], [gv, gr], gamma, lam)
    ev[0].record()
    for i in range(n_ev):
        U.GaeForward([v_d, r_d], [adv], gamma, lam)
        ev[2 * i + 1].record()
        U.GaeBackward([grad_adv], [gv, gr], gamma, lam)
        ev[2 * i + 2].record()
    ev[-1].synchronize()
    t_fwd = sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(n_ev)) / n_ev * 1e-3
    t_bwd = sum(ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(n_ev)) / n_ev * 1e-3
    bytes_launch = 12 * T * B + 4 * B  # either direction: SURVEY.md 8(d)
    dom, t_dom = ("gae_bwd_kernel", t_bwd) if t_bwd >= t_fwd else ("gae_fwd_kernel", t_fwd)

    # ---- both scaling readings in the same line.  Bounded: 3 x 100 steps per leg.
    def leg(Bk, graph, steps=100, warmup=20, rounds=3):
        """`rounds` rounds of `steps` timed steps (barrier-bracketed, max over ranks each); ms_per_step = the MEDIAN round
        (the eager launch path is bimodal from run to run at small B -- 35 vs 63 us per step at B = 8192, the autograd
        engine's cross-thread hand-off -- so every round is listed)."""
        st, keep_alive = make_step(Bk, graph)
        rs = [timed(st, steps, warmup if i == 0 else 0) for i in range(rounds)]
        del keep_alive
        order = sorted(range(rounds), key=lambda i: rs[i][0])
        mx, per = rs[order[rounds // 2]]
        return {"B_per_gpu": Bk, "global_B": Bk * world, "launch": "hipGraph replay" if graph else "eager",
                "ms_per_step": mx / steps * 1e3, "value": T * Bk * world * steps / mx,
                "rounds_ms_per_step": [r[0] / steps * 1e3 for r in rs],
                "per_rank_ms_per_step": [p / steps * 1e3 for p in per]}

    detail = None
    if not args.no_scaling_detail:
        GB = 65536
        detail = {"world_size": world, "backend": (dist.get_backend() if dist is not None else None),
                  "weak": {"eager": leg(B_DEFAULT, False), "graph": leg(B_DEFAULT, True)},
                  "strong": None, "strong_per_rank_probe": None}
        if GB % world == 0:
            detail["strong"] = {"eager": leg(GB // world, False), "graph": leg(GB // world, True)}
        if world == 1:   # what one rank of an N-GPU strong-scaling run of global B = 65536 holds, timed on this GPU
            detail["strong_per_rank_probe"] = {str(n): {"eager": leg(GB // n, False), "graph": leg(GB // n, True)}
                                               for n in (2, 4, 8)}

    # HBM bytes per launch from the PMC counters: they need their own rocprofv3 passes (FETCH_SIZE and WRITE_SIZE do not
    # fit one pass and cannot be combined with tracing), so the figure is the committed result of
    # tests/tools/collect_profiles.sh for this shape (profiles/<round>_gae_pmc_traffic.csv), not a live measurement
    traffic = None
    traffic_source = None
    tj = os.path.join(ROOT, "profiles", "gae_traffic.json")
    if os.path.exists(tj):
        try:
            rec = json.load(open(tj))
            if rec.get("T") == T and rec.get("B") == B:
                traffic = rec.get(dom)
                traffic_source = rec.get("source")
        except Exception:
            traffic = None

    if rank == 0:
        samples = T * B * world
        out = {
            "metric": "gae_fwd_bwd_samples_per_sec",
            "value": samples * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"GAE fwd+bwd, T={T}, B={B} per GPU, fp32 (BASELINE.json configs[1])",
                       "T": T, "B_per_gpu": B, "global_B": global_B,
                       "parallelism": f"batch-sharded x{world}, no data-path collective",
                       "launch": "hipGraph replay" if args.graph else "eager"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": bytes_launch / t_dom / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": bytes_launch / t_dom / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": bytes_launch,
                         "fwd_us": t_fwd * 1e6, "bwd_us": t_bwd * 1e6,
                         "fwd_bwd_frac": (2 * bytes_launch) / (t_fwd + t_bwd) / 1e9 / HBM_PEAK_GBS},
            "per_rank_ms_per_step": [p / args.steps * 1e3 for p in per_rank_s],
            "scaling_detail": detail,
            "cpu_baseline": None if (args.skip_cpu_baseline or world > 1) else cpu_baseline(T, B, gamma, lam),
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
