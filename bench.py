#!/usr/bin/env python3
"""Headline benchmark: GAE forward+backward samples/s at T=1024, B=65536 fp32 (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1: if the process was not started by a launcher (no RANK in the environment) it starts
its own ranks -- it re-executes itself under ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1`` -- so a plain ``python bench.py --gpus 8`` works; under ``torch.distributed.run`` it uses the
ranks it was given.  One rank per GPU over RCCL.

A "step" is one pass of the hot path over one batch of synthetic input: ``adv = GAE(value, reward)`` followed by the
backward pass for both gradients, through the drop-in ``hpc_rll.rl_utils.gae.GAE`` module (HIP kernels behind the C
ABI).  Inputs are resident in HBM before the timed region.  Every trajectory (column) is independent:
the batch axis shards across ranks with NO data-path collective.

What the headline ``value`` is (BASELINE.json metric: "T=1024, B=64k; 1/2/4/8 GPUs"):
  * N = 1: B = 65536 on the one GPU, eager launches (``module(...)`` + ``.backward()``).
  * N > 1: STRONG scaling -- the same GLOBAL batch of 65536 trajectories split over the N ranks (B = 65536/N per GPU),
    ``"scaling": "strong"``, ``"metric_version": 2``.  A rank then holds a step of a few tens of microseconds, the
    launch-latency regime, where the host-side launch path decides -- and which one wins differs from box to box (r03: 35 us
    eager vs 40 us graphed on one, 48 vs 40 on another).  So BOTH are timed in the run (W warmup + K steps each): eager
    ``module(...)`` + ``.backward()``, and ``hpc_rll.graphed`` (forward + backward captured once into a hipGraph, one
    hipGraphLaunch per step; same kernels, same order, same results), and ``hpc_rll.graphed_steps`` (4 steps per hipGraphLaunch:
    two consecutive graph launches leave the GPU idle for 8.5 us, profiles/r05_gae_gaps.txt); the headline is the faster of
    the two ONE-STEP-PER-LAUNCH modes (eager, graph), named in ``config.launch``; graph4 replays one static batch four times per
    launch (a gradient-accumulation reading: the host cannot refresh inputs between its steps), so it is listed in
    ``config.launch_modes`` and never leads.  ``--scaling weak`` / ``--launch eager|graph|graph4`` pin a reading.
    ``scaling_detail.loss_ops`` times what the GAE headline has none of -- the collective: batch-sharded V-trace + TD-lambda
    at the C3 global shape, each forward ending in its ONE all-reduce (RCCL over xGMI), with the all-reduce's share.
Whatever the headline is, the same JSON line carries ``scaling_detail`` with BOTH readings, measured in this run after
the headline region: weak (B = 65536 per GPU) and strong (global B = 65536 split over the N ranks), each eager and as
hipGraph replay, with every rank's step time and the RCCL world size seen.  With one rank it also times the per-rank
shapes an N = 2, 4, 8 strong-scaling run would hold (B = 32768, 16384, 8192).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     -- dominant kernel's algorithmic bytes per launch / its average launch duration measured live: the
                  dispatch's own begin/end timestamps (HIP events attached to the launch through hipExtLaunchKernelGGL
                  on the launch stream = what rocprofv3 --kernel-trace reports) over >= 100 alternating fwd/bwd launches
                  right after the timed region; the hipEventRecord-bracketed figures (they include the gaps between
                  launches) are listed beside them, with the shader clock sampled during the pass.
  cpu_baseline -- oracle/gae_ref.c (a port of the reference algorithm, OpenMP) timed on this host, on a
                  bounded sample of the same workload.  A reported baseline, not the target.  Its key
                  `pytorch_restatement` is the same algorithm run the way hpc_rll.origin runs it (oracle/ref_torch.py:
                  a python loop of fp32 torch CPU ops + autograd backward), on a smaller bounded sample.
  suite        -- (N = 1) BASELINE.json configs[2..4] through the same drop-in modules: V-trace / UPGO / TD-lambda at
                  T=256,B=16384,N=128; PPO at B=65536,N=128; q / C51 / IQN / QR-DQN n-step TD at B=262144 (IQN 65536) -- the
                  reference prints a *_perf() timing for each of them (tests/test_qntd.py:70, test_dntd.py:73,
                  test_qrdqn_nstep_td_error.py:77, test_ppo.py:79); LSTM S=128,B=4096,H=1024; ScatterConnection (cover, add)
                  B=4096,M=256,N=64,64x64 and the packed Pad1D over 2^20 ragged rows: forward / backward ms and roofline
                  fraction each, with `bound` = "hbm", "mfma" or -- for the instruction-bound quantile / C51 forwards --
                  "valu" (bench_suite.py, next to this file, holds the byte / flop / instruction models).
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
    g = torch.Generator().manual_seed(0)
    fp32 = torch.float32
    P = lambda t: ctypes.cast(t.data_ptr(), fp)  # noqa: E731

    def run(Bs, budget, max_reps):
        """fwd + adjoint bwd passes of the C port at batch Bs for ~budget seconds: (samples/s, reps, seconds)."""
        v, r, ga = (torch.randn(T + 1, Bs, generator=g, dtype=fp32), torch.randn(T, Bs, generator=g, dtype=fp32),
                    torch.randn(T, Bs, generator=g, dtype=fp32))
        adv, gv, gr, tab = torch.empty(T, Bs), torch.empty(T + 1, Bs), torch.empty(T, Bs), torch.empty(T)

        def one():
            lib.gae_ref_forward(P(v), P(r), P(adv), T, Bs, gamma, lam)
            lib.gae_ref_backward(P(ga), P(gv), P(gr), P(tab), T, Bs, gamma, lam)

        one()
        reps, t0 = 0, time.perf_counter()
        while True:
            one()
            reps += 1
            dt = time.perf_counter() - t0
            if dt > budget or reps >= max_reps:
                break
        return T * Bs * reps / dt, reps, dt

    # the metric's own configuration (one pass of the C port at the full B is ~0.2 s on this class of
    # host), and the quarter-batch sample of the earlier rounds beside it (columns are independent: same work per sample)
    full, reps, dt = run(B, budget_s * 0.6, 100)
    Bq = max(1, min(B, 16384))
    quarter, reps_q, dt_q = run(Bq, budget_s * 0.4, 200)
    res = {"value": full, "unit": "samples/s", "cores": int(lib.gae_ref_num_threads()), "kind": "port",
           "value_source": "c_port_openmp (oracle/gae_ref.c); the torch restatement, run the way hpc_rll.origin runs, is `pytorch_restatement`",
           "sample": f"T={T} B={B} (the full batch of the metric's configuration) x {reps} fwd+bwd passes, "
                     f"oracle/gae_ref.c OpenMP, {dt:.1f}s",
           "quarter_batch_sample": {"value": quarter, "sample": f"T={T} B={Bq} x {reps_q} passes, {dt_q:.1f}s"}}
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


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this file on this node and hand over."""
    import socket
    one_device = os.environ.get("HPC_RLL_BENCH_ONE_DEVICE") == "1"
    if not one_device and torch.cuda.device_count() < args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus}: this node shows {torch.cuda.device_count()} GPU(s)")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


class SclkSampler:
    """Shader clock during a measurement: the active level of the GPU's pp_dpm_sclk sysfs table, polled from a thread
    (the DVFS state is what makes a profiled and an un-profiled run differ; MI355X_MICROARCH.md)."""

    def __init__(self, index):
        import glob
        self.path, self.samples, self._stop, self._th = None, [], False, None
        cands = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        try:        # the card whose PCI address is this HIP device's (a node exposes all its cards in sysfs)
            pr = torch.cuda.get_device_properties(index)
            want = "%04x:%02x:%02x." % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for c in cands:
                if os.path.basename(os.path.realpath(os.path.dirname(c))).startswith(want):
                    self.path = c
        except Exception:
            self.path = None

    def _read(self):
        try:
            for ln in open(self.path).read().splitlines():
                if ln.rstrip().endswith("*"):
                    return float(ln.split(":")[1].strip().split("M")[0])
        except Exception:
            return None
        return None

    def __enter__(self):
        if self.path is None:
            return self
        import threading

        def run():
            while not self._stop:
                v = self._read()
                if v is not None:
                    self.samples.append(v)
                time.sleep(0.002)
        self._th = threading.Thread(target=run, daemon=True)
        self._th.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._th is not None:
            self._th.join()

    def summary(self):
        if not self.samples:
            return None
        xs = sorted(self.samples)
        return {"median": xs[len(xs) // 2], "min": xs[0], "max": xs[-1], "samples": len(xs), "source": self.path}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # default K: a timed region of >= 2 s at N = 1 (0.244 ms per step), long enough for a 5 s SMI sampler and a DVFS hiccup
    ap.add_argument("--steps", type=int, default=10000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--T", type=int, default=T_DEFAULT)
    ap.add_argument("--B", type=int, default=B_DEFAULT,
                    help="weak scaling: batch per GPU; strong scaling: the GLOBAL batch, split over the ranks")
    ap.add_argument("--skip-cpu-baseline", action="store_true", help="for profiler runs")
    ap.add_argument("--scaling", choices=["auto", "weak", "strong"], default="auto",
                    help="headline reading; auto = strong for N > 1 (global B fixed, SURVEY.md 8d), N = 1 is both")
    ap.add_argument("--launch", choices=["auto", "eager", "graph", "graph4"], default="auto",
                    help="headline launch mode; auto = eager for N = 1, hpc_rll.graphed (hipGraph replay) for N > 1")
    ap.add_argument("--graph", action="store_true", help="same as --launch graph")
    ap.add_argument("--no-scaling-detail", action="store_true", help="skip the extra weak/strong legs (profiler runs)")
    ap.add_argument("--no-suite", action="store_true", help="skip the configs[2..4] suite (N = 1 only)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hpc_rll product path has no CPU fallback")
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)      # does not return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
            # The data path has NO collective (batch-sharded GAE); the process group only carries the barriers and the
            # gather of the per-rank times.  If RCCL cannot come up on a box (IPC mode, container limits) the measurement
            # is still valid over gloo: fall back, and say so in the line (`scaling_detail.backend`).
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                assert int(probe.item()) == world
            except Exception as e:  # noqa: BLE001
                print(f"[bench rank {rank}] RCCL unavailable ({type(e).__name__}: {str(e)[:200]}); control plane over gloo",
                      file=sys.stderr, flush=True)
                try:
                    if dist.is_initialized():
                        dist.destroy_process_group()
                except Exception:  # noqa: BLE001
                    pass
                backend = "gloo"
                dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import hpc_rl_utils as U
    import hpc_rll
    from hpc_rll.rl_utils.gae import GAE
    lib = ctypes.CDLL(os.path.join(ROOT, "di-hpc_amd", "hpc_rll", "_lib", "libhpc_rll_hip.so"))   # already loaded: diagnostics

    scaling = args.scaling if args.scaling != "auto" else ("strong" if world > 1 else "weak")
    launch_arg = "graph" if args.graph else args.launch
    launch = launch_arg
    if launch == "auto":
        launch = "eager"      # N = 1; for N > 1 both modes are timed below and the faster one leads
    T, B, gamma, lam = args.T, args.B, 0.99, 0.97
    global_B = B if scaling == "strong" else B * world
    if scaling == "strong":
        assert B % world == 0, "strong scaling: the global batch must divide by the number of ranks"
        B //= world

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    GRAPH_STEPS = 4   # steps per hipGraphLaunch of the "graph4" launch mode (see make_step)

    def make_step(Bk, graph):
        """(step callable, tensors) for one fwd+bwd pass at batch Bk on this rank; graph=True: hpc_rll.graphed (one
        hipGraphLaunch per step); graph="graph4": hpc_rll.graphed_steps, GRAPH_STEPS steps -- the same kernels, in the same
        order, on the same synthetic batch as every other mode -- per hipGraphLaunch.  Why: the kernel trace of the replay loop
        (profiles/r05_gae_gaps.txt) shows 0.0 us between the two kernels INSIDE a graph launch and 8.5 us of idle GPU between
        two consecutive graph launches (direct launches: 0.0 us both ways, but the module + autograd host path then costs
        ~87 us per step); at B = 8192 per rank that gap is a fifth of the 35 us step.  Several steps per launch pay it once.
        The returned callable runs `n` steps: n // GRAPH_STEPS replays of the multi-step graph, the rest on the one-step one."""
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
        gs = hpc_rll.graphed(gae, value, reward, gamma, lam, grad_outputs=grad_adv)
        if graph != "graph4":
            return gs.replay, (value, reward, grad_adv, gs)
        gm = hpc_rll.graphed_steps(gae, [(value, reward, gamma, lam)] * GRAPH_STEPS, grad_outputs=[grad_adv] * GRAPH_STEPS)

        def run_n(n):
            for _ in range(n // GRAPH_STEPS):
                gm.replay()
            for _ in range(n % GRAPH_STEPS):
                gs.replay()
        run_n.multi = True          # timed() hands it the step count instead of calling it once per step
        return run_n, (value, reward, grad_adv, gs, gm)

    def timed(step, steps, warmup):
        """W untimed steps, barrier + synchronize, EXACTLY `steps` timed steps, barrier + synchronize.
        Returns (max over ranks, per-rank list) of the elapsed seconds."""
        multi = getattr(step, "multi", False)
        if multi:
            step(warmup)
        else:
            for _ in range(warmup):
                step()
        barrier()
        t0 = time.perf_counter()
        if multi:
            step(steps)
        else:
            for _ in range(steps):
                step()
        # closing bracket: this rank's own completion (synchronize), THEN the barrier -- the reported time is the max
        # over ranks of completion times measured from a common start, so a straggler still counts in full, but the
        # latency of the barrier collective itself (tens of us, against K x 40 us at 8 ranks) is not billed to the steps
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        barrier()
        if dist is None:
            return mine, [mine]
        t = torch.tensor([mine], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [x.item() for x in allt]
        return max(per_rank), per_rank

    # device clock / allocator pre-roll: a freshly leased GPU idles at its low power state and the first ~10 ms of
    # work run at ramping clocks.  Untimed, not part of W or K (the W warmup steps and the K timed steps follow).
    n_pre = max(30, int(30 * 65536 / max(B, 1)) if B < 65536 else 30)
    launch_modes = None
    if launch_arg == "auto" and world > 1:
        # N > 1: which host-side launch path is faster at B/N trajectories per rank depends on the box (driver's r03 box at
        # B = 8192: 35.3 us eager vs 39.8 us graphed; builder's: 47.7 vs 40.5) -- so BOTH are timed, W warmup + K steps
        # each, same tensors' shapes, same kernels, and the headline is the faster one, named in config.launch (VERDICT r03
        # item 2a).  Every rank takes the same decision (the times are the max over ranks, gathered).
        launch_modes = {}
        best = None
        for mode in ("eager", "graph", "graph4"):
            st_m, keep_m = make_step(B, {"eager": False, "graph": True}.get(mode, mode))
            if getattr(st_m, "multi", False):
                st_m(n_pre)
            else:
                for _ in range(n_pre):
                    st_m()
            el_m, per_m = timed(st_m, args.steps, args.warmup)
            launch_modes[mode] = {"ms_per_step": el_m / args.steps * 1e3,
                                  "per_rank_ms_per_step": [p_ / args.steps * 1e3 for p_ in per_m]}
            #  graph4 replays 4 steps of ONE static batch per hipGraphLaunch -- the host cannot refresh the inputs
            # between them, which is not the metric's "one training step per launch": it is timed and listed
            # (launch_modes), it never leads.  The headline is the faster of the one-step-per-launch modes.
            if mode != "graph4" and (best is None or el_m < best[0]):
                best = (el_m, per_m, mode, st_m, keep_m)
            del st_m, keep_m
        elapsed, per_rank_s, launch, step, keep = best
        del best
        value, reward, grad_adv = keep[:3]
    else:
        step, keep = make_step(B, {"eager": False, "graph": True}.get(launch, launch))
        value, reward, grad_adv = keep[:3]
        if getattr(step, "multi", False):
            step(n_pre)
        else:
            for _ in range(n_pre):
                step()
        elapsed, per_rank_s = timed(step, args.steps, args.warmup)

    # ---- per-kernel durations, measured IMMEDIATELY after the timed region (same clocks / thermal state), in the same
    # alternating fwd/bwd order as the timed region (a kernel repeated back to back would find its inputs in the
    # 256 MiB Infinity Cache and look faster than it is in the real sequence; rocprofv3 --kernel-trace of this command
    # sees the same pattern).  Two readings of the same >= 100 launch pairs, all on torch's current stream = the launch
    # stream: (a) the dispatch's own begin / end timestamps (hpc_rll_ktime_*: start/stop HIP events attached to the
    # launch itself) -- the kernel's duration as rocprofv3 reports it, used for the roofline figure; (b) hipEventRecord
    # around every launch -- includes the gap to the neighbouring launch, listed for comparison.
    v_d, r_d = value.detach(), reward.detach()
    adv = torch.empty_like(r_d)
    gv, gr = torch.empty_like(v_d), torch.empty_like(r_d)
    n_ev = min(max(args.steps, 100), 500)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n_ev + 1)]
    U.GaeForward([v_d, r_d], [adv], gamma, lam)
    U.GaeBackward([grad_adv], [gv, gr], gamma, lam)
    torch.cuda.synchronize()
    with SclkSampler(dev.index or 0) as sclk:
        assert lib.hpc_rll_ktime_begin(2 * n_ev) == 0
        ev[0].record()
        for i in range(n_ev):
            U.GaeForward([v_d, r_d], [adv], gamma, lam)
            ev[2 * i + 1].record()
            U.GaeBackward([grad_adv], [gv, gr], gamma, lam)
            ev[2 * i + 2].record()
        ev[-1].synchronize()
        kms = (ctypes.c_float * (2 * n_ev))()
        kkind = (ctypes.c_int * (2 * n_ev))()
        n_k = lib.hpc_rll_ktime_end(kms, kkind, 2 * n_ev)
    assert n_k == 2 * n_ev, f"hpc_rll_ktime_end: {n_k}"
    kf = [kms[i] for i in range(n_k) if kkind[i] == 0]
    kb = [kms[i] for i in range(n_k) if kkind[i] == 1]
    t_fwd, t_bwd = sum(kf) / len(kf) * 1e-3, sum(kb) / len(kb) * 1e-3
    t_fwd_ev = sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(n_ev)) / n_ev * 1e-3
    t_bwd_ev = sum(ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(n_ev)) / n_ev * 1e-3
    bytes_launch = 12 * T * B + 4 * B  # either direction: SURVEY.md 8(d)
    dom, t_dom = ("gae_bwd_kernel", t_bwd) if t_bwd >= t_fwd else ("gae_fwd_kernel", t_fwd)
    cfg = {}
    for d, name in ((0, "gae_fwd_kernel"), (1, "gae_bwd_kernel")):
        c6 = (ctypes.c_int * 6)()
        lib.hpc_rll_gae_last_config(d, c6)
        cfg[name] = {"cols_per_lane": c6[0], "steps_per_chunk": c6[1], "waves_per_workgroup": c6[2], "nontemporal": c6[3],
                     "half_wave_tiles": bool(c6[4]), "pipelined": bool(c6[5])}
    del adv, gv, gr

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
        names = {False: "eager", True: "hpc_rll.graphed (hipGraph replay)",
                 "graph4": f"hpc_rll.graphed_steps (hipGraph replay, {GRAPH_STEPS} steps per launch)"}
        return {"B_per_gpu": Bk, "global_B": Bk * world, "launch": names[graph],
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
            detail["strong"] = {"eager": leg(GB // world, False), "graph": leg(GB // world, True), "graph4": leg(GB // world, "graph4")}
        if world == 1:   # what one rank of an N-GPU strong-scaling run of global B = 65536 holds, timed on this GPU
            detail["strong_per_rank_probe"] = {str(n): {"eager": leg(GB // n, False), "graph": leg(GB // n, True),
                                                        "graph4": leg(GB // n, "graph4")} for n in (2, 4, 8)}
            #  the strong-scaling factor an N-rank run of the global batch would read if every rank's step
            # took what ONE GPU measures at B / N here (no collective in the data path; barrier and rank spread not included):
            # this run's own N = 1 step / the per-rank step, per launch mode; `best` = the mode bench.py --gpus N would lead with
            base_ms = elapsed / args.steps * 1e3
            proj = {}
            for n in (2, 4, 8):
                modes = {m: base_ms / detail["strong_per_rank_probe"][str(n)][m]["ms_per_step"] for m in ("eager", "graph", "graph4")}
                modes["best"] = max(modes["eager"], modes["graph"])   # one step per launch only; graph4 listed beside it
                proj[str(n)] = modes
            detail["projected_strong_x"] = {"n1_ms_per_step": base_ms, "by_ranks": proj,
                                            "note": "N = 1 step of this run (B = 65536, eager) / one-GPU step at B = 65536 / N"}

    if detail is not None and world > 1:
        detail["loss_ops"] = loss_ops_leg(dev, dist, world, rank, backend, timed)

    # HBM bytes per launch from the PMC counters: they need their own rocprofv3 passes (FETCH_SIZE and WRITE_SIZE do not
    # fit one pass and cannot be combined with tracing), so the figure is the committed result of
    # tests/tools/collect_profiles.sh for this shape (profiles/<round>_gae_pmc_traffic.csv), not a live measurement --
    # and it is only quoted when the record was taken with the SAME launch configuration the kernels just used.
    traffic = None
    traffic_source = None
    tj = os.path.join(ROOT, "profiles", "gae_traffic.json")
    if os.path.exists(tj):
        try:
            rec = json.load(open(tj))
            if rec.get("T") == T and rec.get("B") == B:
                if rec.get("config") == cfg:
                    traffic = rec.get(dom)
                    traffic_source = rec.get("source")
                else:
                    traffic_source = (f"profiles/gae_traffic.json was recorded with another launch configuration "
                                      f"({rec.get('config')}): not quoted")
        except Exception:
            traffic = None

    suite = None
    if world == 1 and not args.no_suite:
        del step, keep, value, reward, grad_adv, v_d, r_d
        if os.environ.get("HPC_RLL_BENCH_KEEP_CACHE") != "1":
            torch.cuda.empty_cache()
        suite = run_suite(dev)

    if rank == 0:
        out = {
            "metric": "gae_fwd_bwd_samples_per_sec",
            "value": T * global_B * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            # 1 (rounds 1-2): weak scaling, eager launches for every N.  2 (round 3 on): N > 1 is STRONG scaling (global
            # B = 65536 split over the ranks, BASELINE.json's reading) and its launch mode is the faster of eager / hipGraph
            # replay, both timed in the run; N = 1 is unchanged.  The weak / eager reading stays in scaling_detail.
            "metric_version": 2,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"GAE fwd+bwd, T={T}, global B={global_B} ({B} per GPU), fp32 (BASELINE.json configs[1])",
                       "T": T, "B_per_gpu": B, "global_B": global_B,
                       "parallelism": f"batch-sharded x{world}, no data-path collective",
                       "launch": {"graph": "hpc_rll.graphed (hipGraph replay of the same kernels)",
                                  "graph4": f"hpc_rll.graphed_steps (hipGraph replay of the same kernels, {GRAPH_STEPS} steps per launch)"}.get(launch, "eager"),
                       "launch_modes": launch_modes,
                       "backend": (dist.get_backend() if dist is not None else None)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": bytes_launch / t_dom / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": bytes_launch / t_dom / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": bytes_launch,
                         "timing": f"kernel begin/end timestamps of {len(kf)} forward and {len(kb)} backward launches, alternating",
                         "fwd_us": t_fwd * 1e6, "bwd_us": t_bwd * 1e6,
                         "fwd_us_min_max": [min(kf) * 1e3, max(kf) * 1e3], "bwd_us_min_max": [min(kb) * 1e3, max(kb) * 1e3],
                         "fwd_bwd_frac": (2 * bytes_launch) / (t_fwd + t_bwd) / 1e9 / HBM_PEAK_GBS,
                         "frac_rocprof": frac_rocprof(T, B, bytes_launch),
                         "stream_event_fwd_us": t_fwd_ev * 1e6, "stream_event_bwd_us": t_bwd_ev * 1e6,
                         "sclk_mhz": sclk.summary(), "launch_config": cfg},
            "per_rank_ms_per_step": [p / args.steps * 1e3 for p in per_rank_s],
            # what carried the barriers and the gather of the per-rank times: "nccl" = RCCL with `rccl_ranks` ranks; "gloo" =
            # the fall-back (or a test hook) -- then this line is NOT an RCCL result, whatever `n_gpus` says
            "backend": (dist.get_backend() if dist is not None else None),
            "rccl_ranks": (dist.get_world_size() if dist is not None and dist.get_backend() == "nccl" else None),
            "devices_distinct": (not one_device),
            "scaling_detail": detail,
            "suite": suite,
            "cpu_baseline": None if (args.skip_cpu_baseline or world > 1) else cpu_baseline(T, B, gamma, lam),
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def frac_rocprof(T, B, bytes_launch):
    """The same roofline fraction from the COMMITTED rocprofv3 summary (profiles/rNN_gae_bench_kernel_stats.csv, the
    newest round present): kernel-trace average durations of the two GAE kernels of `python bench.py` at this shape.  Lets a
    reader compare the live figure of this box with the committed profile in the line itself; the two
    come from different boxes / runs and differ by the box-to-box spread (a few percent)."""
    import csv
    import glob
    if (T, B) != (T_DEFAULT, B_DEFAULT):
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_gae_bench_kernel_stats.csv")))
    if not files:
        return None
    out = {"source": os.path.relpath(files[-1], ROOT)}
    try:
        for row in csv.DictReader(open(files[-1])):
            name = row["Name"]
            for key, tag in (("gae_fwd", "fwd"), ("gae_bwd", "bwd")):
                if key in name and "coef" not in name and tag + "_us" not in out:
                    avg = float(row["AverageNs"])
                    out[tag + "_us"] = avg / 1e3
                    out[tag + "_calls"] = int(row["Calls"])
                    out[tag] = bytes_launch / (avg * 1e-9) / 1e9 / HBM_PEAK_GBS
        if "fwd" in out and "bwd" in out:
            out["dominant"] = min(out["fwd"], out["bwd"])     # the slower kernel = the lower fraction
            out["fwd_bwd"] = 2 * bytes_launch / ((out["fwd_us"] + out["bwd_us"]) * 1e-6) / 1e9 / HBM_PEAK_GBS
    except Exception as e:  # noqa: BLE001
        out["error"] = repr(e)
    return out


def loss_ops_leg(dev, dist, world, rank, backend, timed, steps=50, warmup=10):
    """the part of north_star the GAE headline does not time -- "a single RCCL all-reduce over xGMI
    for the scalar loss".  V-trace + TD-lambda at the C3 GLOBAL shape (T=256, B=16384, N=128), the batch axis split over the
    ranks, through the drop-in modules with sharded=True: every forward ends in its ONE all-reduce (3 scalars for V-trace,
    1 for TD-lambda; hpc_rll/dist.py), the backward needs none.  A step = V-trace forward + backward, TD-lambda forward +
    backward.  Timed like the headline (barrier + synchronize on both sides, max over ranks).  The all-reduce's share is
    read off a second pass of the SAME local work without the collective (sharded=False and the global loss scale passed
    by hand would need a private entry point, so: the modules as they are, with the process group's all-reduce replaced by
    nothing = sharded=False; the local kernels are identical, the scale differs by a constant factor), plus the bare
    latency of a 3-scalar all-reduce on this group."""
    import time as _t
    from hpc_rll.rl_utils.td import TDLambda
    from hpc_rll.rl_utils.vtrace import VTrace
    T, GB, N = 256, 16384, 128
    if GB % world:
        return {"skipped": f"global B {GB} does not divide by {world} ranks"}
    Bk = GB // world
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    to = torch.randn(T, Bk, N, device=dev, generator=g).requires_grad_(True)
    bo = torch.randn(T, Bk, N, device=dev, generator=g)
    act = torch.randint(0, N, (T, Bk), device=dev, generator=g)
    val = torch.randn(T + 1, Bk, device=dev, generator=g).requires_grad_(True)
    rew = torch.randn(T, Bk, device=dev, generator=g)
    wt = torch.rand(T, Bk, device=dev, generator=g)
    res = {"T": T, "global_B": GB, "B_per_gpu": Bk, "N": N,
           "step": "VTrace fwd (+ its one all-reduce of 3 scalars) + bwd, TDLambda fwd (+ its one all-reduce of 1 scalar) + bwd",
           "backend": dist.get_backend(), "rccl_ranks": dist.get_world_size() if dist.get_backend() == "nccl" else None}
    if dist.get_backend() == "nccl":
        assert dist.get_world_size() == world, "RCCL world size differs from --gpus"
    losses = {}
    for sharded in (True, False):
        vt, td = VTrace(T, Bk, N, sharded=sharded), TDLambda(T, Bk, sharded=sharded)

        def step():
            to.grad = None
            val.grad = None
            ls = vt(to, bo, act, val, rew)
            (ls.policy_loss + ls.value_loss + ls.entropy_loss).sum().backward()
            val.grad = None
            l2 = td(val, rew, wt, 0.9, 0.8)
            l2.sum().backward()
            return ls, l2

        ls, l2 = step()
        losses[sharded] = [float(x.item()) for x in (*ls, l2)]
        mx, per = timed(step, steps, warmup)
        key = "sharded" if sharded else "local_only_no_collective"
        res[key] = {"ms_per_step": mx / steps * 1e3, "per_rank_ms_per_step": [p_ / steps * 1e3 for p_ in per]}
    res["global_losses_vtrace_pg_v_ent_tdlambda"] = losses[True]
    t_s, t_l = res["sharded"]["ms_per_step"], res["local_only_no_collective"]["ms_per_step"]
    res["allreduce_share_of_step"] = max(0.0, (t_s - t_l) / t_s) if t_s > 0 else None
    # bare latency of the collective the forward ends in
    buf = torch.zeros(3, device=dev if backend == "nccl" else "cpu")
    for _ in range(10):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = _t.perf_counter()
    for _ in range(100):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    res["allreduce_3_scalars_us"] = (_t.perf_counter() - t0) / 100 * 1e6
    # every rank must hold the same global loss (the all-reduce's result), bit for bit
    chk = torch.tensor(losses[True], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    res["global_loss_identical_on_every_rank"] = bool(torch.equal(lo, hi))
    return res


def run_suite(dev):
    """BASELINE.json configs[2..4] through the drop-in modules (bounded: well under a minute).  The timing harness and
    the algorithmic byte / flop models are bench_suite.py's (next to this file; the tool DESIGN.md's tables come from)."""
    import bench_suite as S
    S.QUIET = True
    S.dev = dev
    t0 = time.perf_counter()
    out = {}
    for name, fn in (("c3", S.suite_c3), ("ppo", S.suite_ppo), ("td", S.suite_td), ("c4", S.suite_c4),
                     ("c5", lambda: S.suite_c5(quick=True)), ("lstm_mid", S.suite_lstm_mid), ("small", S.suite_small)):
        S.rows.clear()
        try:
            fn()
            for r in S.rows:
                out[r["op"]] = {k: v for k, v in r.items() if k != "op"}
        except Exception as e:      # the headline does not depend on the suite
            out[name + "_error"] = repr(e)
        torch.cuda.empty_cache()
    out["seconds"] = time.perf_counter() - t0
    out["peaks"] = {"hbm_GBs": S.HBM, "mfma_f32_TFLOPs": S.MFMA_F32, "valu_simd_cycles_per_s": S.SIMD_CYCLES}
    return out


if __name__ == "__main__":
    main()
