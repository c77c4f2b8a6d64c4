#!/usr/bin/env python3
"""Suite benchmark at the BASELINE.json configs[2..4] shapes (plus the reference test shapes): per-op forward /
backward time (HIP events on the launch stream, median of interleaved repeats) against the algorithmic-bytes or
flop roofline.  Not the headline bench (bench.py); these are the numbers DESIGN.md quotes.
Writes gpurun_out/suite_<tag>.json and .txt.   Usage: bench_suite.py [c3|td|gemm|c4|c5|small|all]"""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cuda:0")
QUIET = False      # bench.py imports this module for its `suite` object and sets QUIET (one JSON line on stdout)
HBM, MFMA_F32 = 8000.0, 157.3   # GB/s, TFLOP/s (MI355X_MICROARCH.md)
# vector-ALU issue peak: 256 CUs x 4 SIMDs x 2.4 GHz peak shader clock SIMD-cycles per second.  What a wave64 instruction
# costs its SIMD depends on its encoding (tests/tools/micro/valu_rate.hip, profiles/r05_valu_rate.txt: e32 VOP1/VOP2 2.4
# cycles, VOP3 / DPP / v_readlane 4.3, packed fp32 4.4 for two lanes' worth, transcendental 8.2) -- rounds 3-4 priced every
# instruction at 4 cycles, which called a kernel of mostly e32 instructions (C51) "82 % VALU-bound" when removing a fifth of
# its instructions did not change its time.
SIMD_CYCLES = 1024 * 2.4e9
VALU_CPI_DEFAULT = 4.0
rows = []


def valu_counts():
    """Per forward kernel: VALU wave-instructions per SAMPLE from hardware counters (rocprofv3 --pmc SQ_INSTS_VALU,
    tests/tools/r04_td_valu.sh -> profiles/td_valu.json), the mean issue cost of its instructions (static classification of
    the compiled kernel, tests/tools/r05_valu_classify.py -> profiles/r05_td_valu_classes.json) and the hand-counted
    minimum of its inner loop.  Static properties of the compiled kernel at the recorded shape."""
    out = {}
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "td_valu.json")))
        for k, v in rec.get("valu_wave_insts_per_sample", {}).items():
            out[k] = {"insts": v}
    except Exception:  # noqa: BLE001
        return {}
    try:
        cls = json.load(open(os.path.join(ROOT, "profiles", "r05_td_valu_classes.json")))
        for k, v in cls.get("kernels", {}).items():
            if k in out:
                out[k]["cpi"] = v["cycles_per_valu_inst"]
    except Exception:  # noqa: BLE001
        pass
    return out


# Hand counts of the irreducible vector work per SAMPLE, in wave-instructions:
#   QR-DQN / IQN: tau x tau' pairs of (quantile, target), two per packed instruction, NINE packed instructions per two pairs
#                 (e, two clamped differences, two Huber factors, four accumulations), 64 lanes: tau tau' / 2 x 9 / 64, at 4.4 cycles;
#   C51 (one sample per wave): position 10, run table 12, one shuffle step 5, the two fetches and their select 13, log 10,
#                 quotient 7, cross-entropy term and its 64-lane sum 9, per-sample scalars and row addresses 14: 80, at 3.1 cycles.
def valu_min(name, tau=32, tau_p=32):
    if name in ("qrdqn_nstep_td_fwd", "iqn_nstep_td_fwd"):
        return tau * tau_p / 2 * 9 / 64, 4.4
    if name == "dist_nstep_td_fwd":
        return 80.0, 3.1
    return None


# Measured HBM traffic of the suite's kernels: profiles/suite_traffic.json, written by
# tests/tools/collect_suite_pmc.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x 2, calibration copy) at the
# SAME shapes as the rows below.  A static record of a profiler run, like profiles/gae_traffic.json -- the counters cannot be
# collected inside this process.  OP_KERNELS: the kernels of an op's forward / backward (name prefixes as the summary prints them).
OP_KERNELS = {
    "td_lambda": (["colscan_rev_kernel<TdLambdaOp"], ["scale_rows4_kernel"]),
    "vtrace": (["categorical_fwd_kernel<", "categorical_fwd_noent_kernel<", "colscan_rev_kernel<VtraceOp"], ["categorical_bwd_kernel<", "scale_rows4_kernel"]),
    "upgo": (["categorical_fwd_noent_kernel<", "colscan_rev_kernel<UpgoOp"], ["categorical_bwd_kernel<"]),
    "ppo": (["ppo_fwd_fused_kernel<"], None),
    "scatter_cover": (["scatter_out_lds_kernel<false"], ["scatter_bwd_tile_kernel", "scatter_bwd_lds_kernel"]),
    "scatter_add": (["scatter_out_lds_kernel<true"], ["scatter_bwd_tile_kernel", "scatter_bwd_lds_kernel"]),
    "pad1d_packed_api": (["pad1d_packed_wave_kernel", "packed_table_kernel<", "packed_chunk_sums_kernel"], None),
    "dist_nstep_td": (["dist_nstep_fwd_batch_kernel<"], None),
    "iqn_nstep_td": (["iqn_fwd_group_kernel<"], None),
    "qrdqn_nstep_td": (["qrdqn_fwd_quad_kernel<"], None),
}


def traffic_record():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "suite_traffic.json")))
    except Exception:  # noqa: BLE001
        return None


def attach_traffic(row, bytes_f, bytes_b):
    """measured HBM bytes (fetched + written) of the op's forward / backward kernels beside the algorithmic bytes the fraction is
    priced with: `*_traffic_over_algorithmic` well above 1 would mean wasted re-reads"""
    rec = traffic_record()
    if not rec or row["op"] not in OP_KERNELS:
        return
    ks = rec["kernels"]

    def total(prefixes):
        tot, found = 0.0, []
        for pre in prefixes:
            for name, v in ks.items():
                if name.startswith(pre):
                    tot += (v["fetch_mb"] + v["write_mb"]) * 1e6
                    found.append(name)
                    break
        return tot, found

    for tag, prefixes, alg in (("fwd", OP_KERNELS[row["op"]][0], bytes_f), ("bwd", OP_KERNELS[row["op"]][1], bytes_b)):
        if not prefixes or not alg:
            continue
        tot, found = total(prefixes)
        if tot > 0:
            row[tag + "_traffic_bytes"] = tot
            row[tag + "_traffic_over_algorithmic"] = tot / alg
            row[tag + "_traffic_kernels"] = found
    row["traffic_source"] = "profiles/suite_traffic.json (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, tests/tools/collect_suite_pmc.sh)"


def timed(fn, n=5, rounds=3):
    fn()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e-3)
    return statistics.median(ts)


def report(name, shape, t_f, bytes_f, t_b=None, bytes_b=None, flops_f=None, flops_b=None, valu_f=None):
    """valu_f: VALU wave-instructions of one forward launch (hardware counter): when the vector-ALU issue fraction exceeds
    the HBM fraction the forward is instruction-bound and says so -- bound "valu", fwd_frac = the VALU fraction, the HBM
    reading kept as fwd_hbm_frac (an HBM fraction of 0.19 for a kernel whose VALU pipe is 67 % busy
    hides the real fraction)."""
    r = dict(op=name, shape=shape, fwd_ms=t_f * 1e3)
    if flops_f:
        r.update(fwd_tflops=flops_f / t_f / 1e12, fwd_frac=flops_f / t_f / 1e12 / MFMA_F32, bound="mfma")
    else:
        r.update(fwd_gbs=bytes_f / t_f / 1e9, fwd_frac=bytes_f / t_f / 1e9 / HBM, bound="hbm")
        if valu_f and valu_f.get("insts"):
            n, cpi = valu_f["insts"] * valu_f["samples"], valu_f.get("cpi", VALU_CPI_DEFAULT)
            vf = n * cpi / (SIMD_CYCLES * t_f)
            r.update(fwd_valu_frac=vf, fwd_valu_insts=n, fwd_valu_cycles_per_inst=cpi)
            if valu_f.get("min"):
                mn, mcpi = valu_f["min"]
                r.update(fwd_valu_min_insts=mn * valu_f["samples"], fwd_valu_min_frac=mn * valu_f["samples"] * mcpi / (SIMD_CYCLES * t_f))
            if vf > r["fwd_frac"]:
                #  the headline fraction of an instruction-bound forward is the ALGORITHMIC minimum's (hand-counted
                # irreducible instructions x their issue cost / capacity): a kernel that executes more instructions must not
                # score higher.  The executed-count reading (a static record of the compiled kernel, profiles/td_valu.json)
                # stays beside it as fwd_valu_frac.
                head = r.get("fwd_valu_min_frac", vf)
                r.update(bound="valu", fwd_hbm_frac=r["fwd_frac"], fwd_frac=head,
                         bound_note="forward is VALU-issue bound: fwd_frac = hand-counted minimum instructions of the inner loop x cycles per "
                                    "instruction (by encoding, profiles/r05_valu_rate.txt) / (1024 SIMDs x 2.4 GHz x time); fwd_valu_frac = the "
                                    "same with the EXECUTED count (SQ_INSTS_VALU recorded in profiles/td_valu.json for the kernel as compiled "
                                    "then); backward (a write stream) stays HBM")
    if t_b is not None:
        r.update(bwd_ms=t_b * 1e3)
        if flops_b:
            r.update(bwd_tflops=flops_b / t_b / 1e12, bwd_frac=flops_b / t_b / 1e12 / MFMA_F32)
        else:
            r.update(bwd_gbs=bytes_b / t_b / 1e9, bwd_frac=bytes_b / t_b / 1e9 / HBM)
    if not flops_f:
        attach_traffic(r, bytes_f, bytes_b)
    rows.append(r)
    if not QUIET:
        print(json.dumps(r), flush=True)


def fwd_bwd(make_loss, grads_of):
    """time forward alone and backward alone (retain_graph) of a closure returning a scalar-ish loss"""
    loss = make_loss()
    t_f = timed(make_loss)
    loss = make_loss()
    g = torch.ones_like(loss)

    def bwd():
        for p in grads_of:
            p.grad = None
        loss.backward(g, retain_graph=True)

    t_b = timed(bwd)
    return t_f, t_b


def host_path(make_out, grads_of, n=300):
    """Host cost of an eager call at a shape whose kernels take less than the host path: the enqueue-only wall time per call
    (no synchronisation inside the loop; the GPU keeps up) of the forward alone and of forward + backward, and the wall time per
    fwd+bwd call with the final synchronisation included.  host_us = what the CPU spends per call (python + pybind + allocator
    + autograd engine), the quantity a launch-latency-bound training step is made of."""
    import time as _t
    out = make_out()
    g = torch.ones_like(out)

    def fb():
        for p in grads_of:
            p.grad = None
        make_out().backward(g)

    for _ in range(20):
        fb()
    torch.cuda.synchronize()
    res = {}
    for name, fn in (("fwd", make_out), ("fwd_bwd", fb)):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = _t.perf_counter()
            for _ in range(n):
                fn()
            host = (_t.perf_counter() - t0) / n
            torch.cuda.synchronize()
            best = min(best, host)
        res[f"host_us_per_{name}_call"] = best * 1e6
    torch.cuda.synchronize()
    t0 = _t.perf_counter()
    for _ in range(n):
        fb()
    torch.cuda.synchronize()
    res["wall_us_per_fwd_bwd_call"] = (_t.perf_counter() - t0) / n * 1e6
    return res


def timed_graph(fn, n=10, rounds=3):
    """GPU time per call of `fn` with the host taken out: n calls captured into ONE hipGraph, replayed.  For ops whose
    kernels take a few microseconds the eager figures above are the host's (torch's autograd engine: ~27 us per
    backward), not the kernels'."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e-3)
    return statistics.median(ts)


def fwd_bwd_graph(make_loss, grads_of):
    """forward alone, and backward = (forward + backward) - forward, as hipGraph replays (see timed_graph).  The backward
    is captured TOGETHER with its forward: autograd runs a node's backward on the stream its forward ran on, so a
    backward of an eagerly built graph cannot be captured on the capture stream."""
    t_f = timed_graph(make_loss)
    g = torch.ones_like(make_loss())

    def both():
        torch.autograd.grad([make_loss()], grads_of, [g])

    t_fb = timed_graph(both)
    return t_f, max(t_fb - t_f, 1e-9)


def add_kernel_times(t_f, bytes_f, t_b, bytes_b):
    """attach the graph-replay (kernel-bound) readings to the row `report` just appended"""
    rows[-1].update(fwd_kernel_ms=t_f * 1e3, fwd_kernel_frac=bytes_f / t_f / 1e9 / HBM, bwd_kernel_ms=t_b * 1e3,
                    bwd_kernel_frac=bytes_b / t_b / 1e9 / HBM)
    if rows[-1].get("bound") == "valu":
        n_min = rows[-1].get("fwd_valu_min_insts")
        rows[-1].update(fwd_kernel_hbm_frac=rows[-1]["fwd_kernel_frac"],
                        fwd_kernel_valu_frac=rows[-1]["fwd_valu_insts"] * rows[-1]["fwd_valu_cycles_per_inst"] / (SIMD_CYCLES * t_f))
        # same rule as fwd_frac: the minimum's fraction leads (its cost per instruction: valu_min's second value)
        rows[-1]["fwd_kernel_frac"] = (rows[-1]["fwd_valu_min_frac"] * rows[-1]["fwd_ms"] * 1e-3 / t_f) if n_min else rows[-1]["fwd_kernel_valu_frac"]
    if not QUIET:
        print(json.dumps(rows[-1]), flush=True)


def suite_c3(T=256, B=16384, N=128):
    from hpc_rll.rl_utils.td import TDLambda
    from hpc_rll.rl_utils.upgo import UPGO
    from hpc_rll.rl_utils.vtrace import VTrace
    g = torch.Generator(device=dev).manual_seed(0)
    TB = T * B
    value = torch.randn(T + 1, B, device=dev, generator=g, requires_grad=True)
    reward = torch.randn(T, B, device=dev, generator=g)
    weight = torch.rand(T, B, device=dev, generator=g)
    target = torch.randn(T, B, N, device=dev, generator=g, requires_grad=True)
    behaviour = torch.randn(T, B, N, device=dev, generator=g)
    action = torch.randint(0, N, (T, B), device=dev, generator=g)
    rho = torch.rand(T, B, device=dev, generator=g)
    shape = f"T={T} B={B} N={N}"
    m = TDLambda(T, B)
    t_f, t_b = fwd_bwd(lambda: m(value, reward, weight), [value])
    report("td_lambda", shape, t_f, 16 * TB, t_b, 8 * TB)
    add_kernel_times(*[x for pair in zip(fwd_bwd_graph(lambda: m(value, reward, weight), [value]), (16 * TB, 8 * TB)) for x in pair])
    # The torch-side floor of a `.backward()` call on this box, in this process: the smallest possible
    # op through the autograd engine with the same output size -- y = 2 x on a (T+1, B) tensor, backward = ONE elementwise
    # kernel of torch's own (8 T B bytes, the same traffic as TD-lambda's backward kernel).  What TD-lambda's API-level
    # backward takes beyond this is the library's; the rest is the engine handing the graph task to its device thread and
    # back.  The kernel-level reading (hipGraph replay) is listed beside it.
    yy = value * 2.0
    gg = torch.ones_like(yy)

    def floor_bwd():
        value.grad = None
        yy.backward(gg, retain_graph=True)

    t_floor = timed(floor_bwd)
    rows[-1].update(autograd_floor_bwd_ms=t_floor * 1e3, bwd_over_autograd_floor_ms=(t_b - t_floor) * 1e3,
                    autograd_floor_note="y = 2*x on a (T+1,B) tensor, y.backward(): torch's own single elementwise kernel "
                                        "through the same engine, same process, same box")
    if not QUIET:
        print(json.dumps(rows[-1]), flush=True)
    del yy, gg
    m = VTrace(T, B, N)
    # Round 6: the forward is the OP's -- m(...) returning its three losses.  Until round 5 this row timed sum(m(...)): three
    # one-workgroup torch kernels between two forwards, after which the first large kernel runs ~10 % slower (rocprofv3:
    # categorical_fwd_kernel 358 us behind the adds, 326 us behind another large kernel; profiles/r06_cat_pair_probe.txt).  That
    # reading stays in the row as fwd_with_sum_ms.  The backward starts from the three losses directly (one autograd node).
    vfwd = lambda: m(target, behaviour, action, value, reward)  # noqa: E731
    t_f = timed(vfwd)
    t_fs = timed(lambda: sum(vfwd()))
    losses = list(vfwd())
    ones = [torch.ones_like(x) for x in losses]

    def vbwd():
        target.grad = None
        value.grad = None
        torch.autograd.backward(losses, ones, retain_graph=True)

    t_b = timed(vbwd)
    # algorithmic minimum: two logits reads (+ action + O(TB)) forward; logits read + grad write backward
    report("vtrace", shape, t_f, 2 * 4 * TB * N + 8 * TB + 12 * TB, t_b, 2 * 4 * TB * N + 8 * TB)
    rows[-1].update(fwd_with_sum_ms=t_fs * 1e3, fwd_with_sum_frac=(2 * 4 * TB * N + 20 * TB) / t_fs / 1e9 / HBM,
                    fwd_note="fwd_ms = the module's forward (three losses returned); fwd_with_sum_ms = sum() of them inside the timed "
                             "call as rounds 1-5 measured it (three extra one-workgroup torch kernels per forward)")
    del losses, ones
    m = UPGO(T, B, N)
    t_f, t_b = fwd_bwd(lambda: m(target, rho, action, reward, value.detach()), [target])
    report("upgo", shape, t_f, 4 * TB * N + 8 * TB + 16 * TB, t_b, 2 * 4 * TB * N + 8 * TB)


def suite_ppo(B=65536, N=128):
    from hpc_rll.rl_utils.ppo import PPO
    g = torch.Generator(device=dev).manual_seed(0)
    ln = torch.randn(B, N, device=dev, generator=g, requires_grad=True)
    lo = torch.randn(B, N, device=dev, generator=g)
    a = torch.randint(0, N, (B,), device=dev, generator=g)
    vn = torch.randn(B, device=dev, generator=g, requires_grad=True)
    vo, adv, ret = (torch.randn(B, device=dev, generator=g) for _ in range(3))
    import hpc_rl_utils as U
    out5, ws = torch.empty(5, device=dev), U.ppo_workspace(B, dev)
    t_f = timed(lambda: U.PPOForward([ln.detach(), lo, a, vn.detach(), vo, adv, ret, None], [out5, ws], True, 0.2, 0.0))
    g1 = torch.ones(1, device=dev)
    gl, gv = torch.empty(B, N, device=dev), torch.empty(B, device=dev)
    t_b = timed(lambda: U.PPOBackward([g1, g1, g1, ln.detach(), a, ws], [gl, gv]))
    report("ppo", f"B={B} N={N}", t_f, 2 * 4 * B * N + 8 * B + 20 * B, t_b, 2 * 4 * B * N + 8 * B)
    # the host out of the picture (one pybind call per forward is ~20 us of CPU; the kernels take less): hipGraph replays
    t_fk = timed_graph(lambda: U.PPOForward([ln.detach(), lo, a, vn.detach(), vo, adv, ret, None], [out5, ws], True, 0.2, 0.0))
    t_bk = timed_graph(lambda: U.PPOBackward([g1, g1, g1, ln.detach(), a, ws], [gl, gv]))
    add_kernel_times(t_fk, 2 * 4 * B * N + 8 * B + 20 * B, t_bk, 2 * 4 * B * N + 8 * B)


def suite_c4(S=128, B=4096, I=1024, H=1024, L=1):
    from hpc_rll.torch_utils.network.rnn import LSTM
    torch.manual_seed(0)
    m = LSTM(S, B, I, H, L).to(dev)
    x = torch.randn(S, B, I, device=dev, requires_grad=True)
    h0 = torch.randn(L, B, H, device=dev)
    c0 = torch.randn(L, B, H, device=dev)
    flops_f = 2.0 * S * B * 4 * H * (I + H) * L          # x-branch + recurrent GEMM
    flops_b = 2.0 * flops_f                               # dX/dH and dW GEMMs
    y, _ = m(x, (h0, c0))
    t_f = timed(lambda: m(x, (h0, c0)), n=2, rounds=3)
    g = torch.ones_like(y)

    def bwd():
        x.grad = None
        for p in m.parameters():
            p.grad = None
        y.backward(g, retain_graph=True)

    t_b = timed(bwd, n=2, rounds=3)
    report("lstm", f"S={S} B={B} I={I} H={H} L={L}", t_f, None, t_b, None, flops_f, flops_b)


def suite_td(B=1 << 18, N=64, nstep=5, n_atom=51, tau=32):
    """a5-a7 at a batch where the kernels, not the host, set the time.  These ops GATHER the taken action's entry / atom
    row / quantile row per sample (everything else of q is never read) and their backward writes a full one-hot-shaped
    gradient: forward bytes are counted with every gathered piece rounded up to whole 128-byte lines (what HBM has to
    move), backward bytes = the gradient tensor written + the per-sample unit gradient read."""
    from hpc_rll.rl_utils.td import DistNStepTD, IQNNStepTDError, QNStepTD, QRDQNNStepTDError
    g = torch.Generator(device=dev).manual_seed(0)
    line = lambda nbytes: (nbytes + 127) // 128 * 128  # noqa: E731
    act = lambda: torch.randint(0, N, (B,), device=dev, generator=g)  # noqa: E731
    reward = torch.randn(nstep, B, device=dev, generator=g)
    done = (torch.rand(B, device=dev, generator=g) < 0.1).float()
    weight = torch.rand(B, device=dev, generator=g)
    a, na = act(), act()
    per_sample = 16 + 4 * nstep + 4 + 4 + 4   # actions, rewards, done, weight, td_err
    vc = valu_counts()

    def vf(name, samples):
        return dict(vc.get(name, {}), samples=samples, min=valu_min(name, tau, tau))
    # a gathered row that does not start on a 128-byte line touches 1 + (bytes - 4) / 128 lines on average (4-byte aligned start)
    lines_unaligned = lambda nbytes: 128.0 * (1.0 + (nbytes - 4) / 128.0)  # noqa: E731

    q = torch.randn(B, N, device=dev, generator=g, requires_grad=True)
    nq = torch.randn(B, N, device=dev, generator=g)
    m = QNStepTD(nstep, B, N)
    t_f, t_b = fwd_bwd(lambda: m(q, nq, a, na, reward, done, weight, 0.99)[0], [q])
    report("q_nstep_td", f"B={B} N={N} nstep={nstep}", t_f, B * (2 * 128 + per_sample), t_b, B * (4 * N + 4))
    add_kernel_times(*[x for pair in zip(fwd_bwd_graph(lambda: m(q, nq, a, na, reward, done, weight, 0.99)[0], [q]),
                                         (B * (2 * 128 + per_sample), B * (4 * N + 4))) for x in pair])
    del q, nq

    d = torch.softmax(torch.randn(B, N, n_atom, device=dev, generator=g), -1).requires_grad_(True)
    nd = torch.softmax(torch.randn(B, N, n_atom, device=dev, generator=g), -1)
    m = DistNStepTD(nstep, B, N, n_atom)
    t_f, t_b = fwd_bwd(lambda: m(d, nd, a, na, reward, done, weight, 0.99, -10.0, 10.0)[0], [d])
    row_b = line(4 * n_atom) if (4 * n_atom) % 128 == 0 else lines_unaligned(4 * n_atom)
    report("dist_nstep_td", f"B={B} N={N} atoms={n_atom}", t_f, B * (2 * row_b + per_sample + 4 * n_atom),
           t_b, B * (4 * N * n_atom + 4 * n_atom), valu_f=vf("dist_nstep_td_fwd", B))
    rows[-1].update(fwd_line_gather_ceiling_gbs=5800.0, fwd_frac_of_gather_ceiling=rows[-1]["fwd_gbs"] / 5800.0,
                    fwd_bound_note="two 204-byte rows per sample at 4-byte alignment: 2.56 lines of 128 bytes each on average; a bare gather "
                                   "of lines runs at 5.3-5.9 TB/s (profiles/r04_gather_micro.txt)")
    add_kernel_times(*[x for pair in zip(fwd_bwd_graph(lambda: m(d, nd, a, na, reward, done, weight, 0.99, -10.0, 10.0)[0], [d]),
                                         (B * (2 * row_b + per_sample + 4 * n_atom), B * (4 * N * n_atom + 4 * n_atom)))
                       for x in pair])
    del d, nd

    Bi = B // 4
    ai, nai = a[:Bi].contiguous(), na[:Bi].contiguous()
    qi = torch.randn(tau, Bi, N, device=dev, generator=g, requires_grad=True)
    nqi = torch.randn(tau, Bi, N, device=dev, generator=g)
    rq = torch.rand(tau, Bi, device=dev, generator=g)
    m = IQNNStepTDError(tau, tau, nstep, Bi, N)
    t_f, t_b = fwd_bwd(lambda: m(qi, nqi, ai, nai, reward[:, :Bi].contiguous(), done[:Bi].contiguous(), rq, 0.99, 1.0,
                                 weight[:Bi].contiguous())[0], [qi])
    report("iqn_nstep_td", f"tau=tau'={tau} B={Bi} N={N}", t_f, Bi * (2 * tau * 128 + 8 * tau + per_sample), t_b,
           Bi * (4 * tau * N + 4 * tau), valu_f=vf("iqn_nstep_td_fwd", Bi))
    # the forward is a 128-byte-line gather (every other line of q / next_n_q): what that pattern reaches with the loss arithmetic
    # left out is measured (tests/tools/micro/gather.hip, profiles/r04_gather_micro.txt: 5.3-5.9 TB/s of lines in four lane -> row
    # maps, a contiguous read 6.3) -- reported beside the fraction of the 8 TB/s peak
    rows[-1].update(fwd_line_gather_ceiling_gbs=5800.0, fwd_frac_of_gather_ceiling=rows[-1]["fwd_gbs"] / 5800.0,
                    fwd_bound_note="line-granular gather: a bare gather of the same lines runs at 5.3-5.9 TB/s (profiles/r04_gather_micro.txt)")
    ri, di, wi = reward[:, :Bi].contiguous(), done[:Bi].contiguous(), weight[:Bi].contiguous()
    add_kernel_times(*[x for pair in zip(fwd_bwd_graph(lambda: m(qi, nqi, ai, nai, ri, di, rq, 0.99, 1.0, wi)[0], [qi]),
                                         (Bi * (2 * tau * 128 + 8 * tau + per_sample), Bi * (4 * tau * N + 4 * tau))) for x in pair])
    # round 6: the same loss on the quantile-innermost layout (layout='bnt': q (B,N,tau)) -- a sample's quantiles are ONE row
    qb = qi.detach().permute(1, 2, 0).contiguous().requires_grad_(True)
    nqb = nqi.permute(1, 2, 0).contiguous()
    del qi, nqi
    mb = IQNNStepTDError(tau, tau, nstep, Bi, N, layout='bnt')
    t_f, t_b = fwd_bwd(lambda: mb(qb, nqb, ai, nai, ri, di, rq, 0.99, 1.0, wi)[0], [qb])
    by_f, by_b = Bi * (2 * line(4 * tau) + 8 * tau + per_sample), Bi * (4 * tau * N + 4 * tau)
    report("iqn_nstep_td_bnt", f"tau=tau'={tau} B={Bi} N={N}, q (B,N,tau)", t_f, by_f, t_b, by_b)
    rows[-1].update(note="IQNNStepTDError(layout='bnt'), not in the reference: the reference layout's forward is the iqn_nstep_td row")
    add_kernel_times(*[x for pair in zip(fwd_bwd_graph(lambda: mb(qb, nqb, ai, nai, ri, di, rq, 0.99, 1.0, wi)[0], [qb]), (by_f, by_b)) for x in pair])
    del qb, nqb

    qq = torch.randn(B, N, tau, device=dev, generator=g, requires_grad=True)
    nqq = torch.randn(B, N, tau, device=dev, generator=g)
    m = QRDQNNStepTDError(tau, nstep, B, N)
    t_f, t_b = fwd_bwd(lambda: m(qq, nqq, a, na, reward, done, 0.99, weight)[0], [qq])
    report("qrdqn_nstep_td", f"B={B} N={N} tau={tau}", t_f, B * (2 * line(4 * tau) + per_sample + 4 * tau), t_b,
           B * (4 * N * tau + 4 * tau), valu_f=vf("qrdqn_nstep_td_fwd", B))
    add_kernel_times(*[x for pair in zip(fwd_bwd_graph(lambda: m(qq, nqq, a, na, reward, done, 0.99, weight)[0], [qq]),
                                         (B * (2 * line(4 * tau) + per_sample + 4 * tau), B * (4 * N * tau + 4 * tau))) for x in pair])


def suite_gemm():
    import hpc_torch_utils_network as U
    for (M, N, K) in [(4096, 4096, 4096), (4096, 4096, 1024), (524288, 4096, 1024)]:
        a = torch.randn(M, K, device=dev)
        b = torch.randn(K, N, device=dev)
        c = torch.empty(M, N, device=dev)
        t = timed(lambda: U.gemm_f32(a, b, out=c), n=3)
        report("gemm_f32_nn", f"M={M} N={N} K={K}", t, None, flops_f=2.0 * M * N * K)
        bt = torch.randn(N, K, device=dev)          # NT: both operands contiguous along k -> LDS-DMA staging (tune key 25)
        t = timed(lambda: U.gemm_f32(a, bt.t(), out=c), n=3)
        report("gemm_f32_nt", f"M={M} N={N} K={K}", t, None, flops_f=2.0 * M * N * K)
        del a, b, bt, c


def suite_c5(B=4096, M=256, N=64, H=64, W=64, quick=False):
    """quick: Scatter + the packed Pad1D only (bench.py's driver-run `suite`); the list-of-tensors legs build 131k
    tensor objects on the host and take seconds."""
    from hpc_rll.torch_utils.network.scatter_connection import ScatterConnection
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(B, M, N, device=dev, generator=g, requires_grad=True)
    loc = torch.stack([torch.randint(0, H, (B, M), device=dev, generator=g),
                       torch.randint(0, W, (B, M), device=dev, generator=g)], -1)
    for st in ("cover", "add"):
        m = ScatterConnection(B, M, N, H, W, st)
        out = m(x, loc)
        t_f = timed(lambda: m(x, loc), n=3)
        go = torch.randn_like(out)

        def bwd():
            x.grad = None
            out.backward(go, retain_graph=True)

        t_b = timed(bwd, n=3)
        by_f = 4 * B * M * N + 16 * B * M + 4 * B * N * H * W
        # backward: the gather touches essentially every 64-B sector of grad_out at this density
        report(f"scatter_{st}", f"B={B} M={M} N={N} H={H} W={W}", t_f, by_f, t_b, 4 * B * N * H * W + 4 * B * M * N)
        del out, go
    # Pad1D / Unpad1D over n ragged tensors (views of one buffer), len ~ U[32,128)
    from hpc_rll.rl_utils import padding as P
    import numpy as np
    del x, loc
    n1m = 1 << 20
    lens1m = torch.from_numpy(np.random.default_rng(1).integers(32, 128, n1m)).to(dev)
    flat1m = torch.randn(int(lens1m.sum().item()), device=dev)
    t_pk = timed(lambda: P.Padding1DPacked(flat1m, lens1m, max_len=127), n=3)
    report("pad1d_packed_api", f"n={n1m} len~U[32,128) (device table, no host loop)", t_pk, 4 * flat1m.numel() + 8 * n1m * 127)
    del flat1m, lens1m
    if quick:
        return
    import cabi
    n = 1 << 17
    lens = np.random.default_rng(0).integers(32, 128, n)
    flat = torch.randn(int(lens.sum()), device=dev)
    xs = list(torch.split(flat, [int(v) for v in lens]))
    new_x, mask, shapes = P.Padding1D(xs)
    import hpc_rl_utils as U
    table = torch.tensor([[t.data_ptr(), 1, 1, t.shape[0]] for t in xs], dtype=torch.int64).to(dev)
    t_un = timed(lambda: P.UnPadding1D(new_x, shapes), n=1, rounds=2)
    rows.append(dict(op="unpad1d_python_api", shape=f"n={n}", fwd_ms=t_un * 1e3, note="list-of-tensors API incl. host table build"))
    if not QUIET:
        print(json.dumps(rows[-1]), flush=True)
    mx = int(lens.max())
    t_k = timed(lambda: cabi.call("hpc_rll_pad_forward", dev, table.data_ptr(), new_x.data_ptr(), mask.data_ptr(), n, 1, 1, mx, 0), n=5)
    report("pad1d_kernel", f"n={n} len~U[32,128)", t_k, 4 * int(lens.sum()) + 8 * n * mx)
    t_api = timed(lambda: P.Padding1D(xs), n=1, rounds=2)
    rows.append(dict(op="pad1d_python_api", shape=f"n={n}", fwd_ms=t_api * 1e3, note="list-of-tensors API incl. host table build"))
    if not QUIET:
        print(json.dumps(rows[-1]), flush=True)


def suite_small():
    """reference test shapes: launch-latency regime"""
    from hpc_rll.rl_utils.gae import GAE
    from hpc_rll.torch_utils.network.rnn import LSTM
    T, B = 1024, 64
    v = torch.randn(T + 1, B, device=dev, requires_grad=True)
    r = torch.randn(T, B, device=dev)
    m = GAE(T, B)
    t_f, t_b = fwd_bwd(lambda: m(v, r), [v])
    report("gae_small", f"T={T} B={B}", t_f, 12 * T * B, t_b, 12 * T * B)
    rows[-1].update(host_path(lambda: m(v, r), [v]))
    # the same host-side reading for TD-lambda and PPO at shapes whose kernels take a few microseconds
    from hpc_rll.rl_utils.ppo import PPO
    from hpc_rll.rl_utils.td import TDLambda
    Ts, Bs, Ns = 64, 64, 32
    vs = torch.randn(Ts + 1, Bs, device=dev, requires_grad=True)
    rs = torch.randn(Ts, Bs, device=dev)
    td = TDLambda(Ts, Bs)
    rows.append(dict(op="td_lambda_small", shape=f"T={Ts} B={Bs}", **host_path(lambda: td(vs, rs), [vs])))
    ln = torch.randn(Bs, Ns, device=dev, requires_grad=True)
    lo, act = torch.randn(Bs, Ns, device=dev), torch.randint(0, Ns, (Bs,), device=dev)
    vn = torch.randn(Bs, device=dev, requires_grad=True)
    vo, adv, ret = (torch.randn(Bs, device=dev) for _ in range(3))
    ppo = PPO(Bs, Ns, sync_info=False)
    rows.append(dict(op="ppo_small", shape=f"B={Bs} N={Ns}", **host_path(lambda: sum(ppo(ln, lo, act, vn, vo, adv, ret)[0]), [ln, vn])))
    xs = torch.randn(Ts + 1, Bs, device=dev, requires_grad=True)
    rows.append(dict(op="torch_floor_small", shape=f"y = 2 x on ({Ts + 1},{Bs}); y.backward(g)", **host_path(lambda: xs * 2.0, [xs]),
                     note="torch's own smallest op pair through the same autograd engine: the floor of any eager forward + backward here"))
    if not QUIET:
        for r_ in rows[-4:]:
            print(json.dumps(r_), flush=True)
    S, B, I, H, L = 64, 3, 1792, 384, 3
    m = LSTM(S, B, I, H, L).to(dev)
    x = torch.randn(S, B, I, device=dev, requires_grad=True)
    y, _ = m(x, None)
    t_f = timed(lambda: m(x, None), n=2)
    g = torch.ones_like(y)

    def bwd():
        x.grad = None
        y.backward(g, retain_graph=True)

    t_b = timed(bwd, n=2)
    fl = 2.0 * S * B * 4 * H * ((I + H) + 2 * (H + H))
    report("lstm_small", f"S={S} B={B} I={I} H={H} L={L}", t_f, None, t_b, None, fl, 2 * fl)


def suite_lstm_mid():
    """mid-size batches: the persistent mid-batch kernels (csrc/lstm_mid.hpp) at the shapes of the latency
    table, I = H, L = 1, S = 64; us per step = whole forward (backward) / S, beside the matrix floor of the recurrent product."""
    from hpc_rll.torch_utils.network.rnn import LSTM
    import hpc_torch_utils_network as NW
    S = 64
    for B, H in ((64, 1024), (16, 384)):
        torch.manual_seed(0)
        m = LSTM(S, B, H, H, 1).to(dev)
        x = torch.randn(S, B, H, device=dev, requires_grad=True)
        y, _ = m(x, None)
        fpath = NW.lstm_last_forward_path()
        t_f = timed(lambda: m(x, None), n=3)
        g = torch.ones_like(y)

        def bwd():
            x.grad = None
            y.backward(g, retain_graph=True)

        t_b = timed(bwd, n=3)
        fl = 2.0 * S * B * 4 * H * (H + H)
        report("lstm_mid", f"S={S} B={B} I={H} H={H} L=1", t_f, None, t_b, None, fl, 2 * fl)
        rows[-1]["op"] = f"lstm_mid_B{B}_H{H}"
        rows[-1].update(fwd_us_per_step=t_f / S * 1e6, bwd_us_per_step=t_b / S * 1e6, forward_path=fpath,
                        backward_path=NW.lstm_last_backward_path(),
                        recurrent_product_matrix_floor_us=2.0 * B * 4 * H * H / (MFMA_F32 * 1e12) * 1e6)


if __name__ == "__main__":
    import faulthandler
    faulthandler.enable()
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    for _kv in os.environ.get("HPC_RLL_TUNE", "").split(","):      # e.g. HPC_RLL_TUNE=25:0,16:1 (A/B runs of this tool)
        if ":" in _kv:
            import hpc_rl_utils as _U
            _U.tune_set(int(_kv.split(":")[0]), int(_kv.split(":")[1]))
    # steady-state power / clock state: bench.py runs these suites after seconds of GAE work, where the VALU-heavy
    # categorical kernels read ~10 % slower than on a freshly leased GPU (335 -> 365 us per 2.15 GB head; not placement:
    # tests/tools/r03_cat_placement_probe.py shows eight input allocations within 1 % of each other).  Pre-roll with the
    # same work so that this tool and bench.py's `suite` object describe the same state.
    from hpc_rll.rl_utils.gae import GAE as _GAE
    _v = torch.randn(1025, 65536, device=dev, requires_grad=True)
    _r = torch.randn(1024, 65536, device=dev, requires_grad=True)
    _g = torch.randn(1024, 65536, device=dev)
    _m = _GAE(1024, 65536)
    _t0 = __import__("time").time()
    while __import__("time").time() - _t0 < float(os.environ.get("SUITE_PREROLL_S", "4.0")):
        for _ in range(200):
            _v.grad = _r.grad = None
            _m(_v, _r).backward(_g)
        torch.cuda.synchronize()
    del _v, _r, _g, _m
    torch.cuda.empty_cache()
    if which in ("c3", "all"):
        suite_c3()
        suite_ppo()
    if which in ("td", "all"):
        suite_td()
    if which in ("gemm", "all"):
        suite_gemm()
    if which in ("c4", "all"):
        suite_c4()
    if which in ("c5", "all"):
        suite_c5()
    if which in ("small", "all"):
        suite_small()
        suite_lstm_mid()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", f"suite_{which}.json"), "w"), indent=1)
