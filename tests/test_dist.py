"""Batch-axis data parallelism: per-rank sums scaled by 1/(global count) + ONE all-reduce of the loss
scalars reproduce the single-process result; gradients of a shard equal the corresponding rows of the full
gradient; GAE needs no collective.

CPU tier (gloo, world_size 2): the host logic in hpc_rll.dist with the ORACLE as the per-rank op.
GPU tier (2 processes sharing cuda:0, gloo backend on device tensors): the real HIP kernels through the drop-in
modules with ``sharded=True`` (the driver's 8-GPU runs use the same code with backend "nccl" = RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, grad_err, rel_err


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _run_guarded(rank, world, port, q, name, *a):
    """Spawn target (top level: picklable).  Runs the worker `name`; an exception is put on the queue so that the
    parent fails in seconds instead of at its timeout."""
    try:
        globals()[name](rank, world, port, q, *a)
    except BaseException as e:  # noqa: BLE001
        import traceback
        q.put(("error", rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))
        raise


def _collect(q, world, timeout):
    res = []
    for _ in range(world):
        item = q.get(timeout=timeout)
        assert item[0] != "error", f"worker {item[1]} failed:\n{item[2]}"
        res.append(item)
    return sorted(res, key=lambda t: t[0])


def _spawn(target, world, timeout, *args):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_run_guarded, args=(r, world, port, q, target.__name__) + args) for r in range(world)]
    [p.start() for p in ps]
    try:
        return _collect(q, world, timeout)
    finally:
        for p in ps:
            p.join(30)
            if p.is_alive():
                p.kill()


def _data(seed, T, B, N):
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    return dict(value=f(T + 1, B), reward=f(T, B), weight=rng.random((T, B)).astype(np.float32), target=f(T, B, N),
                behaviour=f(T, B, N), action=rng.integers(0, N, (T, B)).astype(np.int64))


# ------------------------------------------------------------------------------------------------ CPU / oracle
def _cpu_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
    from hpc_rll import dist as D
    from oracle import ref_torch as R
    _init(rank, world, port)
    T, B, N = 12, 10, 5
    d = {k: torch.from_numpy(v) for k, v in _data(3, T, B, N).items()}
    sh = {k: D.shard_batch(v, 1, rank, world) for k, v in d.items()}
    # TD-lambda: local mean * (local/global count) == local sum * global scale
    v = sh["value"].double().requires_grad_(True)
    local = R.td_lambda_error(v, sh["reward"].double(), sh["weight"].double(), 0.9, 0.8)
    scale = D.loss_scale(T * (B // world), None, sharded=True)
    contrib = (local * (T * (B // world)) * scale).reshape(1)
    contrib.backward()
    loss = D.all_reduce_losses_(contrib.detach().clone(), None, sharded=True)
    # V-trace: three scalars in one all-reduce
    to = sh["target"].double().requires_grad_(True)
    ls = R.vtrace_error(to, sh["behaviour"].double(), sh["action"], sh["value"].double(), sh["reward"].double(), None)
    three = torch.stack([x.detach() for x in ls]) * (T * (B // world)) * scale
    D.all_reduce_losses_(three, None, sharded=True)
    # PPO-style info slots are means: averaged, not summed
    info = torch.tensor([1.0, 2.0, 3.0, float(rank), float(rank) * 2])
    D.all_reduce_losses_(info, None, sharded=True, mean_slots=(3, 4))
    gathered = D.all_gather_batch(sh["reward"], 1)                 # (T, B/R) shards -> (T, B) on every rank
    assert torch.equal(gathered, d["reward"])
    # bucketed gradient all-reduce (LSTM weight gradients): several buckets, mixed shapes, one None grad
    ps = [torch.nn.Parameter(torch.zeros(s)) for s in [(3, 4), (5,), (2, 2, 2), (7,)]]
    for i, p in enumerate(ps[:3]):
        p.grad = torch.full(p.shape, float(rank + 1) * (i + 1))
    D.all_reduce_grads_(ps, None, bucket_bytes=64)
    assert all(torch.equal(p.grad, torch.full(p.shape, 3.0 * (i + 1))) for i, p in enumerate(ps[:3])) and ps[3].grad is None
    D.all_reduce_grads_(ps, None, average=True)
    assert all(torch.equal(p.grad, torch.full(p.shape, 3.0 * (i + 1))) for i, p in enumerate(ps[:3]))
    # differentiable packed all-reduce (what VTrace / PPO use when sharded): sum, identity backward
    a = torch.tensor([float(rank + 1)], requires_grad=True)
    b = torch.tensor([10.0 * (rank + 1), 7.0], requires_grad=True)
    ra, rb = D.all_reduce_sum((a, b), None, mean_slots=(2,))
    (2 * ra.sum() + 3 * rb.sum()).backward()
    assert ra.item() == 3.0 and rb.tolist() == [30.0, 7.0] and a.grad.item() == 2.0 and b.grad.tolist() == [3.0, 3.0]
    assert D.all_reduce_max_int(5 + 3 * rank) == 8
    q.put((rank, loss.item(), v.grad.numpy(), three.numpy(), info.numpy()))
    dist.destroy_process_group()



def test_cpu_gloo_two_ranks_match_single_process():
    from oracle import ref_torch as R
    world = 2
    res = _spawn(_cpu_worker, world, 120)
    T, B, N = 12, 10, 5
    d = {k: torch.from_numpy(v) for k, v in _data(3, T, B, N).items()}
    v = d["value"].double().requires_grad_(True)
    full = R.td_lambda_error(v, d["reward"].double(), d["weight"].double(), 0.9, 0.8)
    full.backward()
    ls = R.vtrace_error(d["target"].double(), d["behaviour"].double(), d["action"], d["value"].double(), d["reward"].double(), None)
    for rank, loss, gshard, three, info in res:
        assert abs(loss - full.item()) < 1e-12
        k = B // world
        assert np.allclose(gshard, v.grad.numpy()[:, rank * k:(rank + 1) * k], atol=1e-14)
        assert np.allclose(three, [x.item() for x in ls], atol=1e-12)
        assert np.allclose(info, [2.0, 4.0, 6.0, 0.5, 1.0])


# ------------------------------------------------------------------ configs[4]: entity-sharded Scatter + Pad1D/Unpad
def _c5_data(seed=9):
    """n ragged 1-D entity feature lists and a (B,M,N) entity batch with (y,x) cells -- BASELINE.json configs[4] in small."""
    rng = np.random.default_rng(seed)
    n, B, M, N, H, W = 24, 8, 6, 4, 5, 7
    lens = rng.integers(1, 12, n)
    ents = [rng.standard_normal(int(k)).astype(np.float32) for k in lens]
    x = rng.standard_normal((B, M, N)).astype(np.float32)
    loc = np.stack([rng.integers(0, H, (B, M)), rng.integers(0, W, (B, M))], -1).astype(np.int64)
    wmap = rng.standard_normal((B, N, H, W)).astype(np.float32)
    wrow = rng.standard_normal(16).astype(np.float32)              # per-column weights of the padded matrix
    return ents, x, loc, wmap, wrow, (H, W)


def _c5_loss(pad, unpad, scatter, ents, x, loc, wmap, wrow, H, W, width, n_global, b_global, dev="cpu"):
    """The scalar a rank contributes: weighted sums over its shard, already scaled by 1/(GLOBAL counts).
    pad(list, value) -> (new_x, mask, shapes); the padded width is forced to the global `width`."""
    new_x, mask, shapes = pad(ents)
    k = new_x.shape[1]
    assert k <= width
    wr = torch.from_numpy(wrow).to(dev)[:k]
    pad_term = (new_x * mask.to(new_x.dtype) * wr).sum() / n_global
    back = unpad(new_x, shapes)                                   # round trip: bit exact
    assert all(torch.equal(a, b) for a, b in zip(back, ents))
    out = scatter(x, loc)
    sc_term = (out * wmap).sum() / b_global
    return (pad_term + sc_term).reshape(1)


def _cpu_c5_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
    from hpc_rll import dist as D
    from oracle import ref_torch as R
    _init(rank, world, port)
    ents, x, loc, wmap, wrow, (H, W) = _c5_data()
    n, B = len(ents), x.shape[0]
    lo, hi = rank * n // world, (rank + 1) * n // world            # shard by entity index
    mine = [torch.from_numpy(e) for e in ents[lo:hi]]
    width = D.all_reduce_max_int(max(t.numel() for t in mine))     # globally consistent pad width
    xs = D.shard_batch(torch.from_numpy(x), 0, rank, world).requires_grad_(True)
    ls = D.shard_batch(torch.from_numpy(loc), 0, rank, world)
    ws = D.shard_batch(torch.from_numpy(wmap), 0, rank, world)
    loss = _c5_loss(lambda l: R.pad(l, 0), R.unpad, lambda a, b: R.scatter_connection(a, b, H, W, "add"), mine, xs, ls,
                    ws, wrow, H, W, width, n, B)
    loss.backward()
    total = D.all_reduce_losses_(loss.detach().clone(), None, sharded=True)
    q.put((rank, total.item(), width, xs.grad.numpy()))
    dist.destroy_process_group()



def _c5_single(pad, unpad, scatter, dev="cpu"):
    ents, x, loc, wmap, wrow, (H, W) = _c5_data()
    te = [torch.from_numpy(e).to(dev) for e in ents]
    tx = torch.from_numpy(x).to(dev).requires_grad_(True)
    loss = _c5_loss(pad, unpad, scatter, te, tx, torch.from_numpy(loc).to(dev), torch.from_numpy(wmap).to(dev), wrow, H, W,
                    max(len(e) for e in ents), len(ents), x.shape[0], dev)
    loss.backward()
    return loss.item(), max(len(e) for e in ents), tx.grad.cpu().numpy()


@pytest.mark.parametrize("world", [2, 8])
def test_cpu_gloo_configs4_entity_sharded_scatter_pad(world):
    """BASELINE.json configs[4] data-parallel leg on the CPU tier: Pad1D/Unpad + ScatterConnection sharded by entity
    index, scalar loss all-reduced (gloo; world 2 and the configuration's own world 8 -- "sharded over 8 x MI355X") ==
    the single-process loss; the pad width is agreed with one int all-reduce(max); the gradient of a shard equals the
    corresponding rows of the full gradient."""
    from oracle import ref_torch as R
    res = _spawn(_cpu_c5_worker, world, 240)
    full, width, gx = _c5_single(lambda l: R.pad(l, 0), R.unpad, lambda a, b: R.scatter_connection(a, b, 5, 7, "add"))
    k = gx.shape[0] // world
    for rank, total, w, g in res:
        assert abs(total - full) < 1e-5 * max(1.0, abs(full))
        assert w == width
        assert np.allclose(g, gx[rank * k:(rank + 1) * k], atol=1e-7)


def test_shard_and_scale_helpers():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
    from hpc_rll import dist as D
    x = torch.arange(24).reshape(2, 12)
    assert torch.equal(D.shard_batch(x, 1, 1, 3), x[:, 4:8])
    assert D.shard_batch(None, 1, 0, 2) is None
    assert D.loss_scale(10) == 0.1 and D.world_size() == 1
    t = torch.ones(3)
    assert D.all_reduce_losses_(t, None, sharded=False) is t


# ------------------------------------------------------------------------------------------------ GPU / HIP kernels
def _gpu_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
    from hpc_rll import dist as D
    from hpc_rll.rl_utils.gae import GAE
    from hpc_rll.rl_utils.td import TDLambda
    from hpc_rll.rl_utils.vtrace import VTrace
    _init(rank, world, port)
    dev = torch.device("cuda:0")
    T, B, N = 40, 256, 12
    d = {k: torch.from_numpy(v) for k, v in _data(4, T, B, N).items()}
    sh = {k: D.shard_batch(v, 1, rank, world).to(dev) for k, v in d.items()}
    v = sh["value"].clone().requires_grad_(True)
    to = sh["target"].clone().requires_grad_(True)
    l1 = TDLambda(T, B // world, sharded=True)(v, sh["reward"], sh["weight"])
    l3 = VTrace(T, B // world, N, sharded=True)(to, sh["behaviour"], sh["action"], v, sh["reward"])
    (l1 + sum(l3)).sum().backward()
    adv = GAE(T, B // world)(sh["value"], sh["reward"])          # no collective
    # batch-sharded LSTM: activations stay local, the weight gradients are summed in one bucketed all-reduce
    from hpc_rll.torch_utils.network.rnn import LSTM
    S, LB, I, H, L = 6, 8, 10, 12, 2
    torch.manual_seed(11)
    m = LSTM(S, LB // world, I, H, L).to(dev)
    x = torch.randn(S, LB, I, generator=torch.Generator().manual_seed(5))
    xs = D.shard_batch(x, 1, rank, world).to(dev).requires_grad_(True)
    y, _ = m(xs, None)
    y.sum().backward()
    D.all_reduce_grads_(list(m.parameters()), None)
    lstm = [p.grad.cpu().numpy() for p in m.parameters()] + [xs.grad.cpu().numpy()]
    # f-2: a batch-sharded per-sample output replicated on every rank, on DEVICE tensors
    adv_full = D.all_gather_batch(adv, 1)
    # configs[4]: entity-sharded Pad1D/Unpad + ScatterConnection through the HIP kernels, scalar loss all-reduced
    from hpc_rll.rl_utils.padding import Padding1D, UnPadding1D
    from hpc_rll.torch_utils.network.scatter_connection import ScatterConnection
    ents, ex, loc, wmap, wrow, (H, W) = _c5_data()
    n, EB = len(ents), ex.shape[0]
    lo, hi = rank * n // world, (rank + 1) * n // world
    mine = [torch.from_numpy(e).to(dev) for e in ents[lo:hi]]
    width = D.all_reduce_max_int(max(t.numel() for t in mine), device=dev)
    exs = D.shard_batch(torch.from_numpy(ex), 0, rank, world).to(dev).requires_grad_(True)
    sc = ScatterConnection(EB // world, ex.shape[1], ex.shape[2], H, W, "add")
    c5 = _c5_loss(lambda l: Padding1D(l, value=0), UnPadding1D, sc, mine, exs,
                  D.shard_batch(torch.from_numpy(loc), 0, rank, world).to(dev),
                  D.shard_batch(torch.from_numpy(wmap), 0, rank, world).to(dev), wrow, H, W, width, n, EB, dev)
    c5.backward()
    c5_total = D.all_reduce_losses_(c5.detach().clone(), None, sharded=True)
    q.put((rank, l1.item(), [x.item() for x in l3], v.grad.cpu().numpy(), to.grad.cpu().numpy(), adv.cpu().numpy(), lstm,
           adv_full.cpu().numpy(), (c5_total.item(), width, exs.grad.cpu().numpy())))
    dist.destroy_process_group()



@pytest.mark.gpu
def test_gpu_two_ranks_match_single_process():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
    from hpc_rll.rl_utils.gae import GAE
    from hpc_rll.rl_utils.td import TDLambda
    from hpc_rll.rl_utils.vtrace import VTrace
    world = 2
    res = _spawn(_gpu_worker, world, 300)
    dev = torch.device("cuda:0")
    T, B, N = 40, 256, 12
    d = {k: torch.from_numpy(v).to(dev) for k, v in _data(4, T, B, N).items()}
    v = d["value"].clone().requires_grad_(True)
    to = d["target"].clone().requires_grad_(True)
    l1 = TDLambda(T, B)(v, d["reward"], d["weight"])
    l3 = VTrace(T, B, N)(to, d["behaviour"], d["action"], v, d["reward"])
    (l1 + sum(l3)).sum().backward()
    adv = GAE(T, B)(d["value"], d["reward"]).cpu().numpy()
    from hpc_rll.torch_utils.network.rnn import LSTM
    S, LB, I, H, L = 6, 8, 10, 12, 2
    torch.manual_seed(11)
    m = LSTM(S, LB, I, H, L).to(dev)
    x = torch.randn(S, LB, I, generator=torch.Generator().manual_seed(5)).to(dev).requires_grad_(True)
    y, _ = m(x, None)
    y.sum().backward()
    full = [p.grad.cpu().numpy() for p in m.parameters()]
    k = B // world
    from hpc_rll.rl_utils.padding import Padding1D, UnPadding1D
    from hpc_rll.torch_utils.network.scatter_connection import ScatterConnection
    ex_shape = _c5_data()[1].shape
    c5_full, c5_width, c5_gx = _c5_single(lambda l: Padding1D(l, value=0), UnPadding1D,
                                          ScatterConnection(ex_shape[0], ex_shape[1], ex_shape[2], 5, 7, "add"), dev)
    for rank, r1, r3, gv, gt, radv, lstm, adv_full, (c5_total, c5_w, c5_g) in res:
        assert np.array_equal(adv_full, adv)                       # all_gather_batch on device tensors: bit equal
        ke = c5_gx.shape[0] // world
        assert abs(c5_total - c5_full) < 1e-5 * max(1.0, abs(c5_full)) and c5_w == c5_width
        assert np.array_equal(c5_g, c5_gx[rank * ke:(rank + 1) * ke])   # scatter backward is a gather: bit equal
        # Summed weight gradients == full-batch gradients.  The half batches (B=4) run the layer-wavefront kernels and
        # the full batch (B=8) the step kernels, so the two sides differ by fp32 rounding through S*L LayerNorms over
        # H=12 columns (tests/tools/lstm_dp_err_probe.py: 1e-5 and 4e-5 from the fp64 oracle respectively); the bound is
        # the gradient tolerance test_lstm_oracle uses.
        for a, b in zip(full, lstm[:-1]):
            assert rel_err(a, b) < 2e-4
        kb = LB // world
        assert grad_err(x.grad.cpu().numpy()[:, rank * kb:(rank + 1) * kb], lstm[-1]) < 2e-4
        assert rel_err(l1.item(), r1) < 1e-6
        assert rel_err([x.item() for x in l3], r3) < 1e-6
        assert grad_err(v.grad.cpu().numpy()[:, rank * k:(rank + 1) * k], gv) < 1e-6
        assert grad_err(to.grad.cpu().numpy()[:, rank * k:(rank + 1) * k], gt) < 1e-6
        assert np.array_equal(adv[:, rank * k:(rank + 1) * k], radv)     # GAE columns are independent: bit equal


def _rccl_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
    from hpc_rll import dist as D
    from hpc_rll.rl_utils.gae import GAE
    from hpc_rll.rl_utils.ppo import PPO
    from hpc_rll.rl_utils.vtrace import VTrace
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    T, B, N = 20, 64, 6
    d = {k: torch.from_numpy(v).to(dev) for k, v in _data(6, T, B, N).items()}
    to = d["target"].clone().requires_grad_(True)
    v = d["value"].clone().requires_grad_(True)
    l3 = VTrace(T, B, N, sharded=True)(to, d["behaviour"], d["action"], v, d["reward"])
    sum(l3).sum().backward()
    adv = GAE(T, B)(d["value"], d["reward"])
    full = D.all_gather_batch(adv, 1)                              # all_gather_into_tensor over RCCL on device tensors
    width = D.all_reduce_max_int(17)                               # device picked from the backend (nccl -> current GPU)
    rng = np.random.default_rng(1)
    f = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(dev)  # noqa: E731
    ln = f(B, N).requires_grad_(True)
    act = torch.from_numpy(rng.integers(0, N, B)).to(dev)
    loss, info = PPO(B, N, sharded=True)(ln, f(B, N), act, f(B), f(B), f(B), f(B))
    sum(loss).sum().backward()
    q.put((rank, [x.item() for x in l3], to.grad.cpu().numpy(), bool(torch.equal(full, adv)), width,
           [x.item() for x in loss], list(info), ln.grad.cpu().numpy()))
    dist.destroy_process_group()



@pytest.mark.gpu
def test_gpu_rccl_backend_on_device_tensors():
    """The `nccl` (= RCCL) backend with device tensors through every hpc_rll.dist entry point the sharded modules use:
    packed differentiable all-reduce (VTrace, PPO), all_gather_batch, the int all-reduce(max).  One rank (a second
    rank needs a second GPU; the multi-rank arithmetic is covered by the gloo tests above): results must equal the
    unsharded modules exactly."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
    from hpc_rll.rl_utils.ppo import PPO
    from hpc_rll.rl_utils.vtrace import VTrace
    (res,) = _spawn(_rccl_worker, 1, 300)
    _, r3, gt, gathered_ok, width, rl, rinfo, gln = res
    dev = torch.device("cuda:0")
    T, B, N = 20, 64, 6
    d = {k: torch.from_numpy(v).to(dev) for k, v in _data(6, T, B, N).items()}
    to = d["target"].clone().requires_grad_(True)
    v = d["value"].clone().requires_grad_(True)
    l3 = VTrace(T, B, N)(to, d["behaviour"], d["action"], v, d["reward"])
    sum(l3).sum().backward()
    assert [x.item() for x in l3] == r3 and np.array_equal(to.grad.cpu().numpy(), gt)
    assert gathered_ok and width == 17
    rng = np.random.default_rng(1)
    f = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(dev)  # noqa: E731
    ln = f(B, N).requires_grad_(True)
    act = torch.from_numpy(rng.integers(0, N, B)).to(dev)
    loss, info = PPO(B, N)(ln, f(B, N), act, f(B), f(B), f(B), f(B))
    sum(loss).sum().backward()
    assert [x.item() for x in loss] == rl and list(info) == rinfo and np.array_equal(ln.grad.cpu().numpy(), gln)


def _check_bench_line(d, scaling, launch_word, B_per_gpu, world=2, steps=5, warmup=2):
    assert d["n_gpus"] == world and d["steps"] == steps and d["warmup"] == warmup and d["scaling"] == scaling
    assert d["config"]["B_per_gpu"] == B_per_gpu and d["config"]["global_B"] == world * B_per_gpu
    assert d["config"]["backend"] == "gloo" and d["metric_version"] == 2
    #  what carried the run is at the TOP level: a gloo line cannot be read as an RCCL result
    assert d["backend"] == "gloo" and d["rccl_ranks"] is None and d["devices_distinct"] is False
    if launch_word is not None:
        assert launch_word in d["config"]["launch"] and d["config"]["launch_modes"] is None
    else:      # auto for N > 1: the host launch paths are all timed in the run; the faster ONE-STEP-PER-LAUNCH mode leads (ADVICE
        lm = d["config"]["launch_modes"]     # r05: graph4 replays one static batch four times per launch -- listed, never the headline)
        assert set(lm) == {"eager", "graph", "graph4"} and all(len(v["per_rank_ms_per_step"]) == world for v in lm.values())
        fastest = min(("eager", "graph"), key=lambda k: lm[k]["ms_per_step"])
        assert {"graph": "hpc_rll.graphed ("}.get(fastest, "eager") in d["config"]["launch"]
        assert "graphed_steps" not in d["config"]["launch"]
        assert abs(d["ms_per_step"] - lm[fastest]["ms_per_step"]) < 1e-9
    assert len(d["per_rank_ms_per_step"]) == world and d["cpu_baseline"] is None and d["suite"] is None
    assert abs(d["value"] - 1024 * world * B_per_gpu / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    rf = d["roofline"]
    assert rf["fwd_us"] > 0 and rf["bwd_us"] > 0 and rf["stream_event_fwd_us"] >= 0.8 * rf["fwd_us"]
    assert rf["launch_config"]["gae_fwd_kernel"]["cols_per_lane"] in (1, 2, 4)
    sd = d["scaling_detail"]
    assert sd["world_size"] == world and sd["backend"] == "gloo"
    for reading, per_gpu in (("weak", 65536), ("strong", 65536 // world)):
        for launch in ("eager", "graph"):
            leg = sd[reading][launch]
            assert leg["B_per_gpu"] == per_gpu and leg["global_B"] == world * per_gpu
            assert len(leg["per_rank_ms_per_step"]) == world and len(leg["rounds_ms_per_step"]) == 3 and leg["ms_per_step"] > 0
    assert sd["strong_per_rank_probe"] is None        # only printed by a single rank
    lo = sd["loss_ops"]      # V-trace + TD-lambda at the C3 global shape, batch-sharded, ONE all-reduce per forward in the step
    assert lo["global_B"] == 16384 and lo["B_per_gpu"] == 16384 // world and lo["backend"] == "gloo" and lo["rccl_ranks"] is None
    assert lo["sharded"]["ms_per_step"] > 0 and lo["local_only_no_collective"]["ms_per_step"] > 0
    assert len(lo["sharded"]["per_rank_ms_per_step"]) == world and 0.0 <= lo["allreduce_share_of_step"] < 1.0
    assert lo["allreduce_3_scalars_us"] > 0 and lo["global_loss_identical_on_every_rank"] is True
    assert len(lo["global_losses_vtrace_pg_v_ent_tdlambda"]) == 4


def _bench_env():
    return dict(os.environ, HPC_RLL_BENCH_ONE_DEVICE="1", HPC_RLL_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")


@pytest.mark.gpu
def test_bench_two_ranks_emits_both_scaling_readings(tmp_path):
    """bench.py under torch.distributed.run with TWO ranks (the driver's launch form): one JSON line from rank 0 with the
    headline, per-rank step times, and `scaling_detail` carrying weak AND strong readings (eager and hipGraph) with the
    world size it saw.  Two ranks share the one GPU of the test box, so the test hooks put both on cuda:0 over gloo
    (RCCL refuses two ranks per device); the driver's 2/4/8-GPU runs use the same code with one rank per GPU over RCCL.
    Here with the weak reading / eager launches selected explicitly."""
    import json
    import subprocess
    import sys
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--B", "4096", "--skip-cpu-baseline", "--scaling", "weak", "--launch", "eager"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=_bench_env(), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    _check_bench_line(json.loads(lines[0]), "weak", "eager", 4096)


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """a plain `python bench.py --gpus 2` (no launcher -- the form of the driver's N=1 command) starts
    its own two ranks and exits 0.  Defaults for N > 1: the STRONG reading (--B is the global batch, split over the
    ranks); eager launches AND hpc_rll.graphed replay are both timed and the faster leads; the
    `scaling_detail.loss_ops` leg times batch-sharded V-trace + TD-lambda with their one all-reduce in the step (2b)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in _bench_env().items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--B", "8192",
           "--skip-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    _check_bench_line(json.loads(lines[0]), "strong", None, 4096)


@pytest.mark.gpu
def test_bench_eight_ranks_on_one_device():
    """the code path `bench.py --gpus 8` takes on the driver's 8-GPU node, run here with its EIGHT ranks
    sharing the one GPU of the test box (the test hooks: every rank on cuda:0, control plane over gloo -- RCCL refuses two
    ranks per device).  The default form: no launcher, strong scaling (global B split 8 ways), all three launch modes
    timed, the faster one-step-per-launch mode leading; `scaling_detail` with the weak / strong legs of eight ranks and the
    loss-ops leg with its all-reduce; the top-level `backend` says what carried it."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in _bench_env().items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2", "--B", "8192",
           "--skip-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    _check_bench_line(json.loads(lines[0]), "strong", None, 1024, world=8)


def _rccl_multi_worker(rank, world, port, q):
    """One rank per GPU over RCCL (runs only on a box with >= 2 GPUs)."""
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
    from hpc_rll import dist as D
    from hpc_rll.rl_utils.gae import GAE
    from hpc_rll.rl_utils.vtrace import VTrace
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    try:          # an RCCL that cannot come up on this box (container limits, IPC mode) is an environment problem: reported
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)     # as a skip, not as a parity failure
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        assert probe.item() == world
    except Exception as e:  # noqa: BLE001
        q.put((rank, "rccl_unavailable", f"{type(e).__name__}: {e}"))
        return
    T, B, N = 20, 64 * world, 6
    d = {k: torch.from_numpy(v) for k, v in _data(6, T, B, N).items()}
    sh = {k: D.shard_batch(v, 1, rank, world).to(dev) for k, v in d.items()}
    to = sh["target"].clone().requires_grad_(True)
    v = sh["value"].clone().requires_grad_(True)
    l3 = VTrace(T, B // world, N, sharded=True)(to, sh["behaviour"], sh["action"], v, sh["reward"])
    sum(l3).sum().backward()
    adv = GAE(T, B // world)(sh["value"], sh["reward"])
    full = D.all_gather_batch(adv, 1)
    s3 = D.all_reduce_sum([torch.full((1,), float(rank + 1), device=dev), torch.full((2,), 2.0 * (rank + 1), device=dev)])
    width = D.all_reduce_max_int(10 + rank)
    p = torch.nn.Parameter(torch.zeros(5, 3, device=dev))
    p.grad = torch.full((5, 3), float(rank + 1), device=dev)
    p2 = torch.nn.Parameter(torch.zeros(7, device=dev))
    p2.grad = torch.arange(7, device=dev, dtype=torch.float32) * (rank + 1)
    D.all_reduce_grads_([p, p2])
    q.put((rank, [x.item() for x in l3], to.grad.cpu().numpy(), v.grad.cpu().numpy(), full.cpu().numpy(),
           [x.cpu().numpy() for x in s3], width, p.grad.cpu().numpy(), p2.grad.cpu().numpy()))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_multi_gpu_matches_single_process():
    """RCCL with MORE THAN ONE rank (VERDICT r02 missing #6): runs whenever the box has >= 2 GPUs, one rank per GPU,
    through every hpc_rll.dist entry point -- all_reduce_sum (packed differentiable all-reduce inside sharded V-trace
    and directly), all_gather_batch, all_reduce_max_int, all_reduce_grads_ -- and compares with the single-process
    result on GPU 0."""
    ngpu = torch.cuda.device_count()
    if ngpu < 2:
        pytest.skip(f"needs >= 2 GPUs for one RCCL rank per device (this box has {ngpu})")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
    from hpc_rll.rl_utils.gae import GAE
    from hpc_rll.rl_utils.vtrace import VTrace
    world = min(ngpu, 8)
    res = sorted(_spawn(_rccl_multi_worker, world, 600), key=lambda r: r[0])
    if any(len(r) > 1 and isinstance(r[1], str) and r[1] == "rccl_unavailable" for r in res):
        pytest.skip("RCCL could not be initialised on this box: " + "; ".join(str(r[2]) for r in res if r[1] == "rccl_unavailable")[:500])
    dev = torch.device("cuda:0")
    T, B, N = 20, 64 * world, 6
    d = {k: torch.from_numpy(v).to(dev) for k, v in _data(6, T, B, N).items()}
    to = d["target"].clone().requires_grad_(True)
    v = d["value"].clone().requires_grad_(True)
    l3 = VTrace(T, B, N)(to, d["behaviour"], d["action"], v, d["reward"])
    sum(l3).sum().backward()
    adv = GAE(T, B)(d["value"], d["reward"]).cpu().numpy()
    k = B // world
    tri = world * (world + 1) / 2
    for rank, r3, gt, gvv, full, s3, width, pg, p2g in res:
        assert rel_err([x.item() for x in l3], r3) < 1e-6
        assert grad_err(to.grad.cpu().numpy()[:, rank * k:(rank + 1) * k], gt) < 1e-6
        assert grad_err(v.grad.cpu().numpy()[:, rank * k:(rank + 1) * k], gvv) < 1e-6
        assert np.array_equal(full, adv)
        assert s3[0].tolist() == [tri] and s3[1].tolist() == [2 * tri, 2 * tri]
        assert width == 10 + world - 1
        assert np.array_equal(pg, np.full((5, 3), tri, np.float32))
        assert np.array_equal(p2g, np.arange(7, dtype=np.float32) * tri)
