"""GPU parity tests for Pad/Unpad (1-3D, grouped) and ScatterConnection: integer / index work, BIT EXACT
(``torch.equal``) against the golden fixtures and the oracle."""
import random

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import ref_torch as R

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


# ------------------------------------------------------------------------------------------------ padding
def test_padding_golden_bit_exact(golden):
    from hpc_rll.rl_utils import padding as P
    g = golden("padding")
    for nd, pad, unpad in ((1, P.Padding1D, P.UnPadding1D), (2, P.Padding2D, P.UnPadding2D), (3, P.Padding3D, P.UnPadding3D)):
        xs = [G(g[f"d{nd}_x{j}"]) for j in range(int(g[f"d{nd}_n"]))]
        for value in (0, -3):
            new_x, mask, shapes = pad(xs, value=value)
            assert mask.dtype == torch.int32 and new_x.dtype == torch.float32
            assert np.array_equal(new_x.cpu().numpy(), g[f"d{nd}_v{value}_new_x"])
            assert np.array_equal(mask.cpu().numpy(), g[f"d{nd}_v{value}_mask"].astype(np.int32))
            assert shapes == [int(v) for v in g[f"d{nd}_shapes"].reshape(-1)]
        back = unpad(new_x, shapes)
        assert len(back) == len(xs)
        for a, b in zip(xs, back):
            assert a.shape == b.shape and torch.equal(a, b)


def _ragged(rng, n, rank):
    # reference ranges (tests/test_padding.py:10-13): 1-D [32,128), 2-D [48,80)x[32,64), 3-D [24,32)x[24,32)x[32,40)
    lo_hi = {1: [(32, 128)], 2: [(48, 80), (32, 64)], 3: [(24, 32), (24, 32), (32, 40)]}[rank]
    return [torch.from_numpy(rng.standard_normal([int(rng.integers(lo, hi)) for lo, hi in lo_hi]).astype(np.float32)).to(DEV)
            for _ in range(n)]


@pytest.mark.parametrize("rank", [1, 2, 3])
def test_padding_round_trip_reference_shapes(rank):
    from hpc_rll.rl_utils import padding as P
    pad = {1: P.Padding1D, 2: P.Padding2D, 3: P.Padding3D}[rank]
    unpad = {1: P.UnPadding1D, 2: P.UnPadding2D, 3: P.UnPadding3D}[rank]
    rng = np.random.default_rng(rank)
    xs = _ragged(rng, 64, rank)
    new_x, mask, shapes = pad(xs)
    ox, om, _ = R.pad([t.cpu() for t in xs], 0)
    assert torch.equal(new_x.cpu(), ox) and torch.equal(mask.cpu(), om.to(torch.int32))
    for a, b in zip(xs, unpad(new_x, shapes)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("rank,mode", [(1, "oracle"), (1, "sample"), (2, "oracle"), (2, "sample"), (3, "oracle"), (3, "sample")])
def test_group_padding_round_trip(rank, mode):
    """tests/test_padding.py:58-99: grouped padding returns per-group tensors; unpadding the groups returns the inputs
    sorted by element count, bit exact; the oracle grouping never pads more than the single-group layout."""
    from hpc_rll.rl_utils import padding as P
    pad = {1: P.Padding1D, 2: P.Padding2D, 3: P.Padding3D}[rank]
    unpad = {1: P.UnPadding1D, 2: P.UnPadding2D, 3: P.UnPadding3D}[rank]
    rng = np.random.default_rng(10 + rank)
    random.seed(rank)
    xs = _ragged(rng, 64, rank)
    group = 4
    new_x, mask, shapes = pad(xs, group=group, group_mode=mode)
    assert isinstance(new_x, tuple) and len(new_x) == len(mask) == len(shapes) <= group
    srt = sorted(xs, key=lambda t: t.numel())
    back = unpad(new_x, shapes)
    assert len(back) == len(srt)
    for a, b in zip(srt, back):
        assert a.shape == b.shape and torch.equal(a, b)
    for x, m in zip(new_x, mask):
        assert x.shape == m.shape and m.dtype == torch.int32
    if mode == "oracle":
        single = pad(xs)[0].numel()
        assert sum(x.numel() for x in new_x) <= single
        if rank == 1:
            pos = R.oracle_split_group([t.numel() for t in srt], group)
            assert [x.shape[0] for x in new_x] == [pos[i + 1] - pos[i] for i in range(group)]


def test_padding_many_small_tensors():
    """C5-style: many entities in one launch (n = 20000, len in [32,128))."""
    from hpc_rll.rl_utils import padding as P
    rng = np.random.default_rng(0)
    lens = rng.integers(32, 128, 20000)
    flat = torch.from_numpy(rng.standard_normal(int(lens.sum())).astype(np.float32)).to(DEV)
    xs = list(torch.split(flat, [int(v) for v in lens]))
    new_x, mask, shapes = P.Padding1D(xs, value=7)
    assert new_x.shape == (20000, int(lens.max()))
    assert int(mask.eq(1).sum()) == int(lens.sum())
    assert bool(new_x[mask.eq(7)].eq(7.0).all())
    back = P.UnPadding1D(new_x, shapes)
    assert torch.equal(torch.cat(back), flat)


def test_packed_padding_matches_list_api_at_one_million_entities():
    """configs[4] scale: n = 2^20 entities in ONE launch, table built on the device (no per-tensor host work)."""
    from hpc_rll.rl_utils import padding as P
    n = 1 << 20
    rng = np.random.default_rng(5)
    lens = torch.from_numpy(rng.integers(32, 128, n)).to(DEV)
    flat = torch.randn(int(lens.sum().item()), device=DEV)
    new_x, mask = P.Padding1DPacked(flat, lens, max_len=127, value=-2)
    assert new_x.shape == (n, 127) and int(mask.eq(1).sum()) == flat.numel()
    assert bool(new_x[mask.eq(-2)].eq(-2.0).all())
    assert torch.equal(P.UnPadding1DPacked(new_x, lens), flat)
    # agreement with the reference-style list API on a prefix
    k = 4096
    xs = list(torch.split(flat[: int(lens[:k].sum().item())], [int(v) for v in lens[:k].tolist()]))
    lx, lm, _ = P.Padding1D(xs, value=-2)
    w = lx.shape[1]
    assert torch.equal(new_x[:k, :w], lx) and torch.equal(mask[:k, :w], lm)


def _packed_group_reference(flat, lens, group, value):
    """What hpc_rll/rl_utils/padding.py:20-45 does, restated for packed rows with torch ops: stable sort by length, the
    split policy on the sorted lengths (the C ABI's host entry point, itself pinned to the reference DP by
    tests/test_host_logic.py), one dense pad per group."""
    import ctypes
    import cabi as N
    lens_c = lens.cpu().numpy()
    order = np.argsort(lens_c, kind="stable")
    sl = lens_c[order]
    n = len(sl)
    sizes = (ctypes.c_int32 * n)(*[int(v) for v in sl])
    shapes = (ctypes.c_int32 * group)()
    pos = (ctypes.c_int32 * (group + 1))()
    ng = N.lib.hpc_rll_oracle_split_group(sizes, n, 1, group, shapes, pos)
    assert ng >= 1
    offs = np.concatenate([[0], np.cumsum(lens_c)])
    fc = flat.cpu()
    xs, ms = [], []
    for g in range(ng):
        rows = order[pos[g]:pos[g + 1]]
        w = int(shapes[g])
        x = torch.full((len(rows), w), float(value))
        m = torch.zeros((len(rows), w), dtype=torch.int32)
        for r, i in enumerate(rows):
            L = int(lens_c[i])
            x[r, :L] = fc[offs[i]:offs[i] + L]
            m[r, :L] = 1
        xs.append(x)
        ms.append(m)
    return xs, ms, [sl[pos[g]:pos[g + 1]] for g in range(ng)], order, list(pos[:ng + 1])


@pytest.mark.parametrize("n,lo,hi,group", [(1, 3, 4, 4), (2, 1, 2, 2), (7, 5, 6, 3), (300, 0, 9, 4), (1000, 32, 128, 8), (5000, 1, 4, 6),
                                           (4097, 200, 1000, 5), (2500, 0, 300, 16), (3000, 7, 8, 2), (1025, 1, 3, 63)])
def test_packed_group_padding_on_device_matches_the_reference_policy(n, lo, hi, group):
    """f-3 on the device: histogram -> split on runs -> stable radix sort -> one pad launch.  Bit
    exact against the reference policy incl. its tie rule: few distinct lengths (cuts forced inside runs), a single
    length, group > n, zero-length rows, max_len >= 256 (two radix passes), ragged last sort chunk."""
    from hpc_rll.rl_utils import padding as P
    rng = np.random.default_rng(n * 31 + group)
    lens = torch.from_numpy(rng.integers(lo, hi, n)).to(DEV)
    flat = torch.randn(int(lens.sum().item()) + 1, device=DEV)[: int(lens.sum().item())]
    for value, max_len in ((0, None), (-4, hi + 5)):
        xs, ms, ls, order = P.Padding1DPacked(flat, lens, max_len=max_len, value=value, group=group, group_mode="oracle")
        rx, rm, rl, rorder, _ = _packed_group_reference(flat, lens, group, value)
        assert np.array_equal(order.cpu().numpy(), rorder)
        assert len(xs) == len(rx) == len(ms) == len(ls)
        for a, b, c, d, e, f in zip(xs, rx, ms, rm, ls, rl):
            assert a.shape == b.shape and torch.equal(a.cpu(), b) and torch.equal(c.cpu(), d)
            assert np.array_equal(e.cpu().numpy(), f)


def test_packed_group_padding_equals_host_dp_at_200k_and_runs_at_one_million():
    """n = 2 x 10^5: cuts, order and widths identical to the host DP.  n = 2^20, group = 8 (configs[4] scale): the same
    checks against the host policy evaluated on the sorted lengths, every row spot-checked through its mask, and the
    whole call (plan + sort + one host sync + pad of ~0.8 GB of output) inside 5 ms."""
    import time
    from hpc_rll.rl_utils import padding as P
    for n, timed in ((200000, False), (1 << 20, True)):
        rng = np.random.default_rng(n)
        lens = torch.from_numpy(rng.integers(32, 128, n)).to(DEV)
        total = int(lens.sum().item())
        flat = torch.randn(total, device=DEV)
        xs, ms, ls, order = P.Padding1DPacked(flat, lens, max_len=127, group=8)
        lens_c = lens.cpu().numpy()
        rorder = np.argsort(lens_c, kind="stable")
        assert np.array_equal(order.cpu().numpy(), rorder)
        import ctypes
        import cabi as N
        sl = lens_c[rorder]
        sizes = (ctypes.c_int32 * n)(*sl.tolist())
        shapes, pos = (ctypes.c_int32 * 8)(), (ctypes.c_int32 * 9)()
        assert N.lib.hpc_rll_oracle_split_group(sizes, n, 1, 8, shapes, pos) == 8
        assert [x.shape[0] for x in xs] == [pos[g + 1] - pos[g] for g in range(8)]
        assert [x.shape[1] for x in xs] == list(shapes)
        offs = torch.cat([torch.zeros(1, dtype=torch.int64, device=DEV), lens.cumsum(0)])
        for g in range(8):
            rows = order[pos[g]:pos[g + 1]]
            assert torch.equal(ms[g].sum(1).long(), lens[rows]) and torch.equal(ls[g], lens[rows])
            assert bool((ms[g][:, 1:] <= ms[g][:, :-1]).all())                      # a mask row is 1...10...0
            # the values: row r of the group, column c < len  ==  flat[offs[row] + c]
            w = xs[g].shape[1]
            col = torch.arange(w, device=DEV)[None, :]
            src = (offs[rows][:, None] + col).clamp(max=total - 1)
            want = torch.where(ms[g].bool(), flat[src], torch.zeros((), device=DEV))
            assert torch.equal(xs[g], want)
        if timed:
            best = 1e9
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = P.Padding1DPacked(flat, lens, max_len=127, group=8)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            del out
            assert best < 3.5e-3, f"grouped packed pad of 2^20 rows took {best * 1e3:.2f} ms"   # measured 1.16 ms end to end (r04), incl. the one host sync


def test_packed_group_padding_round_trip():
    """Padding1DPacked(group > 1) followed by UnPadding1DPackedGrouped returns the flat values in their original order."""
    from hpc_rll.rl_utils import padding as P
    for n, lo, hi, group, mode in ((5000, 32, 128, 8, "oracle"), (333, 0, 7, 4, "oracle"), (70000, 1, 300, 12, "sample"), (1, 4, 5, 3, "oracle")):
        rng = np.random.default_rng(n + group)
        lens = torch.from_numpy(rng.integers(lo, hi, n)).to(DEV)
        flat = torch.randn(int(lens.sum().item()), device=DEV)
        xs, ms, ls, order = P.Padding1DPacked(flat, lens, group=group, group_mode=mode, seed=3)
        back = P.UnPadding1DPackedGrouped(xs, ls, order, total=flat.numel())
        assert torch.equal(back, flat), (n, group, mode)


def test_packed_group_padding_sample_policy_and_errors():
    from hpc_rll.rl_utils import padding as P
    rng = np.random.default_rng(3)
    n = 5000
    lens = torch.from_numpy(rng.integers(10, 90, n)).to(DEV)
    flat = torch.randn(int(lens.sum().item()), device=DEV)
    a = P.Padding1DPacked(flat, lens, group=5, group_mode="sample", seed=11)
    b = P.Padding1DPacked(flat, lens, group=5, group_mode="sample", seed=11)
    assert len(a[0]) <= 5 and all(torch.equal(x, y) for x, y in zip(a[0], b[0]))
    order = a[3]
    sl = lens[order]
    assert bool((sl[1:] >= sl[:-1]).all()) and sum(x.shape[0] for x in a[0]) == n
    k = 0
    for x, m, l in zip(*a[:3]):
        assert x.shape[1] == int(l.max()) and torch.equal(l, sl[k:k + x.shape[0]]) and torch.equal(m.sum(1).long(), l)
        k += x.shape[0]
    widths = [x.shape[1] for x in a[0]]
    assert all(p < q for p, q in zip(widths, widths[1:]))              # equal-width neighbours are merged
    with pytest.raises(RuntimeError):
        P.Padding1DPacked(flat, lens, max_len=50, group=4)               # a length beyond max_len
    with pytest.raises(RuntimeError):
        P.Padding1DPacked(flat, lens, max_len=20000, group=4)            # beyond the histogram width of the device split


@pytest.mark.parametrize("n,lo,hi,max_len", [(5000, 32, 128, 127), (3001, 1, 9, 8), (777, 0, 5, 6), (100, 900, 1100, 1100), (4096, 16, 17, 16),
                                             (1, 5, 6, 5), (70000, 0, 3, 4),
                                             # round 4, wave tiles of 1024 output elements (32 <= max_len <= 16384): the narrowest
                                             # rows (34 per tile), empty rows, widths around the tile size, the widest rows
                                             (20011, 0, 33, 32), (9000, 20, 64, 63), (3000, 100, 300, 299), (700, 1000, 1030, 1029),
                                             (333, 0, 1025, 1024), (40, 5000, 16385, 16384), (1, 40, 41, 40), (7, 0, 1, 64)])
def test_packed_padding_lds_kernels_bit_exact(n, lo, hi, max_len):
    """The LDS-staged packed kernels (round 3: 16 rows per workgroup; round 4: pad on wave tiles in output space): ragged
    spans that start / end off a 16-byte boundary, rows of one element, empty rows, rows wider than a workgroup's worth,
    an unaligned `flat` view; pad and unpad bit exact against a torch restatement, round trip exact."""
    from hpc_rll.rl_utils import padding as P
    rng = np.random.default_rng(n + max_len)
    lens = torch.from_numpy(rng.integers(lo, hi, n)).to(DEV)
    total = int(lens.sum().item())
    for shift in (0, 1, 3):
        flat = torch.randn(total + 8, device=DEV)[shift:shift + total]
        x, m = P.Padding1DPacked(flat, lens, max_len=max_len, value=-7)
        offs = torch.cat([torch.zeros(1, dtype=torch.int64, device=DEV), lens.cumsum(0)])[:-1]
        col = torch.arange(max_len, device=DEV)[None, :]
        valid = col < lens[:, None]
        src = (offs[:, None] + col).clamp(max=max(total - 1, 0))
        want = torch.where(valid, flat[src] if total else torch.zeros(n, max_len, device=DEV), torch.full((), -7.0, device=DEV))
        assert torch.equal(x, want) and torch.equal(m, torch.where(valid, 1, -7).to(torch.int32))
        back = P.UnPadding1DPacked(x, lens, total=total)
        assert torch.equal(back, flat)
        out = torch.full((total + 8,), 3.0, device=DEV)                  # unpad into an unaligned view: neighbours untouched
        import hpc_rl_utils as U
        assert torch.equal(U.unpad1d_packed(x, lens, total), flat)
        del out


def test_packed_padding_wave_tiles_equal_workgroup_kernel():
    """hpc_rll_tune_set key 28: the round-4 pad kernel (wave tiles of 1024 consecutive output elements) against the round-3
    one (16 rows per workgroup) on the configs[4] row distribution, bit for bit, values and mask."""
    import hpc_rl_utils as U
    from hpc_rll.rl_utils import padding as P
    n = 1 << 18
    lens = torch.from_numpy(np.random.default_rng(3).integers(32, 128, n)).to(DEV)
    flat = torch.randn(int(lens.sum().item()), device=DEV)
    try:
        U.tune_set(28, 0)
        x0, m0 = P.Padding1DPacked(flat, lens, max_len=127, value=2)
        U.tune_set(28, 1)
        x1, m1 = P.Padding1DPacked(flat, lens, max_len=127, value=2)
    finally:
        U.tune_set(28, 1)
    assert torch.equal(x0, x1) and torch.equal(m0, m1)


def test_packed_padding_survives_a_violated_precondition():
    """lengths beyond max_len are the caller's error (validate=True reports it); unchecked, the kernels must still stay
    inside their buffers: rows are truncated to max_len columns, nothing crashes, the next call works."""
    from hpc_rll.rl_utils import padding as P
    rng = np.random.default_rng(2)
    n = 3000
    lens = torch.from_numpy(rng.integers(50, 400, n)).to(DEV)
    flat = torch.randn(int(lens.sum().item()), device=DEV)
    x, m = P.Padding1DPacked(flat, lens, max_len=100)
    torch.cuda.synchronize()
    offs = torch.cat([torch.zeros(1, dtype=torch.int64, device=DEV), lens.cumsum(0)])[:-1]
    col = torch.arange(100, device=DEV)[None, :]
    valid = col < lens[:, None]
    assert torch.equal(x, torch.where(valid, flat[(offs[:, None] + col).clamp(max=flat.numel() - 1)], torch.zeros((), device=DEV)))
    assert torch.equal(m, valid.to(torch.int32))
    with pytest.raises(ValueError):
        P.Padding1DPacked(flat, lens, max_len=100, validate=True)
    P.UnPadding1DPacked(x, lens, total=flat.numel())                     # reads only inside x
    torch.cuda.synchronize()


def test_padding_errors():
    from hpc_rll.rl_utils import padding as P
    with pytest.raises(RuntimeError):
        P.Padding1D([torch.zeros(3), torch.zeros(4)])            # host tensors
    with pytest.raises(RuntimeError):
        P.Padding1D([torch.zeros(3, device=DEV), torch.zeros(4, 2, device=DEV)])   # mixed rank
    x = torch.zeros(2, 5, device=DEV)
    with pytest.raises(RuntimeError):
        P.UnPadding1D(x, [3, 9])                                  # does not fit


# ------------------------------------------------------------------------------------------------ scatter
def test_scatter_golden(golden):
    from hpc_rll.torch_utils.network.scatter_connection import ScatterConnection
    g = golden("scatter")
    for i, (B, M, N, H, W, _) in enumerate(g["cases"]):
        B, M, N, H, W = int(B), int(M), int(N), int(H), int(W)
        for st in ("add", "cover"):
            x = G(g[f"c{i}_x"]).requires_grad_(True)
            out = ScatterConnection(B, M, N, H, W, st)(x, G(g[f"c{i}_location"]))
            (out * out).mean().backward()
            assert np.array_equal(out.detach().cpu().numpy(), g[f"c{i}_{st}_out"]), (i, st)   # bit exact
            assert rel_err(g[f"c{i}_{st}_grad_x"], x.grad.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("B,M,N,H,W", [(256, 256, 256, 16, 16), (8, 300, 7, 5, 9), (3, 50, 130, 64, 64), (2, 17, 4, 200, 200)])
def test_scatter_oracle_bit_exact(B, M, N, H, W):
    """tests/test_scatter.py:11-15 shape (B=M=N=256, H=W=16: heavy collisions) and odd shapes.  Both modes are
    compared bit-for-bit with the CPU oracle (sequential scatter_/scatter_add_), as the reference test does."""
    from hpc_rll.torch_utils.network.scatter_connection import ScatterConnection
    rng = np.random.default_rng(B + M)
    x = rng.standard_normal((B, M, N)).astype(np.float32)
    loc = np.stack([rng.integers(0, H, (B, M)), rng.integers(0, W, (B, M))], -1).astype(np.int64)
    gout = rng.standard_normal((B, N, H, W)).astype(np.float32)
    for st in ("cover", "add"):
        xo = torch.from_numpy(x).requires_grad_(True)
        oo = R.scatter_connection(xo, torch.from_numpy(loc), H, W, st)
        oo.backward(torch.from_numpy(gout))
        xd = G(x).requires_grad_(True)
        od = ScatterConnection(B, M, N, H, W, st)(xd, G(loc))
        od.backward(G(gout))
        assert torch.equal(od.detach().cpu(), oo.detach()), st
        assert torch.equal(xd.grad.cpu(), xo.grad), st


@pytest.mark.parametrize("B,M,N,H,W", [(70, 200, 64, 64, 64), (300, 256, 16, 32, 64), (257, 37, 17, 64, 64), (1100, 1, 4, 64, 32)])
def test_scatter_backward_against_cpu_gather(B, M, N, H, W):
    """The backward (one workgroup per (batch element, channel group) staging whole planes in LDS) against the CPU gather:
    more batch elements than CUs, planes of 2048 and 4096 elements, odd N, M = 1, out-of-range locations (gradient 0).
    (Until round 4 this compared the persistent software-pipelined experiment of tune key 34 -- slower, removed in round 5.)"""
    import hpc_torch_utils_network as NW
    rng = np.random.default_rng(B + N)
    loc = np.stack([rng.integers(-1, H + 1, (B, M)), rng.integers(-1, W + 1, (B, M))], -1).astype(np.int64)
    gout = torch.from_numpy(rng.standard_normal((B, N, H, W)).astype(np.float32)).to(DEV)
    dloc = torch.from_numpy(loc).to(DEV)
    res = {}
    for key in (0, 1):          # twice: into NaN-filled buffers, same bits
        gx = torch.full((B, M, N), float("nan"), device=DEV)
        NW.ScatterConnectionBackward([gout, dloc], [gx])
        res[key] = gx.clone()
    assert torch.equal(res[0], res[1])
    y, xx = loc[..., 0], loc[..., 1]
    ok = (y >= 0) & (y < H) & (xx >= 0) & (xx < W)
    gc = gout.cpu().numpy()
    want = np.zeros((B, M, N), np.float32)
    bb, mm = np.nonzero(ok)
    want[bb, mm, :] = gc[bb, :, y[bb, mm], xx[bb, mm]]
    assert np.array_equal(res[1].cpu().numpy(), want)


def test_scatter_deterministic_and_full_write():
    """`cover` is deterministic (the reference kernel is a race) and the output needs no pre-zeroing."""
    import hpc_torch_utils_network as U
    rng = np.random.default_rng(1)
    B, M, N, H, W = 4, 64, 8, 4, 4
    x = G(rng.standard_normal((B, M, N)).astype(np.float32))
    loc = G(np.stack([rng.integers(0, H, (B, M)), rng.integers(0, W, (B, M))], -1).astype(np.int64))
    outs = []
    for fill in (float("nan"), 123.0):
        out = torch.full((B, N, H, W), fill, device=DEV)
        U.ScatterConnectionForward([x, loc], [out], "cover")
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])
    assert not torch.isnan(outs[0]).any()


@pytest.mark.parametrize("B,M,N,H,W", [(3, 3000, 8, 32, 32), (2, 1030, 6, 10, 6), (5, 257, 12, 16, 16), (2, 64, 70, 8, 8),
                                       (1, 9000, 4, 64, 64)])
def test_scatter_many_entities_and_both_forward_kernels(B, M, N, H, W):
    """(a) the index kernel links the per-cell chains chunk by chunk, so M is no longer bounded by
    LDS or an O(M^2) scan -- thousands of entities per map, heavy collisions; (b) the LDS-staged streaming forward
    kernel (tune key 17 = 1, several channel-group widths) and the round-1 kernel (key 17 = 0) both equal the CPU
    oracle bit for bit, cover and add, incl. out-of-range locations, M % 4 != 0, N % 4 != 0."""
    import hpc_torch_utils_network as NW
    from oracle import ref_torch as R
    rng = np.random.default_rng(M + N)
    x = rng.standard_normal((B, M, N)).astype(np.float32)
    loc = np.stack([rng.integers(0, H, (B, M)), rng.integers(0, W, (B, M))], -1).astype(np.int64)
    dx, dloc = torch.from_numpy(x).to(DEV), torch.from_numpy(loc).to(DEV)
    try:
        for typ in ("cover", "add"):
            ref = R.scatter_connection(torch.from_numpy(x), torch.from_numpy(loc), H, W, typ).numpy()
            for lds, npb in ((0, 0), (1, 0), (1, 4), (1, 16)):
                NW.tune_set(17, lds)
                NW.tune_set(18, npb)
                out = torch.full((B, N, H, W), float("nan"), device=DEV)
                NW.ScatterConnectionForward([dx, dloc], [out], typ)
                assert np.array_equal(out.cpu().numpy(), ref), (typ, lds, npb)
    finally:
        NW.tune_set(17, 1)
        NW.tune_set(18, 0)


@pytest.mark.parametrize("n", [1, 7, 2048, 2049, 70001, 2048 * 1024 + 5])
@pytest.mark.parametrize("misalign", [0, 1])
def test_packed_table_rows_through_the_c_abi(n, misalign):
    """hpc_rll_packed_table writes {base + stride * (exclusive prefix of the lengths), 1, 1, length} per row (round 4: rows staged
    through LDS, a wave's stores contiguous; 16-byte stores when the table is 16-byte aligned, 8-byte ones otherwise).  Against
    numpy's cumsum, bit for bit: one chunk, ragged last chunks, more than 1024 chunks (the separate scan launch), zero and large
    lengths, a table at 8 mod 16."""
    import cabi as N
    rng = np.random.default_rng(n)
    lens = rng.integers(0, 200, size=n).astype(np.int64)
    if n > 3:
        lens[n // 2] = 0
        lens[n // 3] = 1 << 33
    base, stride = 1 << 40, 4
    d_lens = torch.from_numpy(lens).to(DEV)
    buf = torch.full((4 * n + 2,), -7, dtype=torch.int64, device=DEV)
    table = buf[misalign:misalign + 4 * n]
    assert table.data_ptr() % 16 == 8 * misalign
    scratch = torch.empty(int(N.lib.hpc_rll_packed_table_scratch_int64(n)), dtype=torch.int64, device=DEV)
    rc = N.lib.hpc_rll_packed_table(d_lens.data_ptr(), n, base, stride, table.data_ptr(), scratch.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    got = buf.cpu().numpy()
    want = np.empty((n, 4), dtype=np.int64)
    want[:, 0] = base + stride * (np.cumsum(lens) - lens)
    want[:, 1] = 1
    want[:, 2] = 1
    want[:, 3] = lens
    assert np.array_equal(got[misalign:misalign + 4 * n].reshape(n, 4), want)
    assert got[:misalign].tolist() == [-7] * misalign and got[misalign + 4 * n:].tolist() == [-7] * (2 - misalign)


@pytest.mark.parametrize("B,M,N,H,W", [(3, 256, 64, 64, 64), (2, 500, 12, 32, 32), (2, 512, 8, 16, 16), (2, 513, 8, 16, 16),
                                       (1, 3000, 8, 32, 32), (4, 1, 4, 8, 8), (3, 37, 17, 16, 12), (2, 255, 70, 4, 4), (2, 1024, 8, 32, 32),
                                       (2, 1025, 8, 32, 32), (3, 700, 5, 16, 16), (2, 200, 64, 128, 128), (2, 300, 8, 100, 100),
                                       (6, 256, 8, 32, 32), (5, 240, 33, 32, 40), (4, 120, 64, 64, 64)])
def test_scatter_in_kernel_index_build(B, M, N, H, W):
    """Round 4 (tune key 37): the LDS-staged forward kernel builds the owner table (cover: LDS atomic max; add: LDS atomic min for the
    head + a broadcast search for the next entity of the chain, M <= 512) itself instead of reading the index launch's tables.
    Against the CPU oracle and against key 37 = 0, bit for bit: heavy collisions (4 x 4 maps), M = 1, M % 4 != 0, the add
    fallback above 512 entities, cover with several entities per thread, and -- on / off only, the oracle has no such case --
    out-of-range locations (dropped).  Round 5: `add` gives a cell with several entities a row of its own in the staged tile (the
    chain's sum) for up to 32 such cells per batch element and walks the chains in the stream loop beyond that: 256 entities on a
    32 x 32 map hold 27 such cells on average -- batch elements on both sides of the limit -- 240 on 32 x 40 about 20, 120 on
    64 x 64 two."""
    import hpc_torch_utils_network as NW
    from oracle import ref_torch as R
    rng = np.random.default_rng(M * 7 + N)
    x = rng.standard_normal((B, M, N)).astype(np.float32)
    loc = np.stack([rng.integers(0, H, (B, M)), rng.integers(0, W, (B, M))], -1).astype(np.int64)
    bad = loc.copy()
    bad[:, ::5, 0] = -1
    bad[:, 1::7, 1] = W
    bad[:, 2::11, 0] = H + 3
    dx = torch.from_numpy(x).to(DEV)
    try:
        for typ in ("cover", "add"):
            ref = R.scatter_connection(torch.from_numpy(x), torch.from_numpy(loc), H, W, typ).numpy()
            for l, want in ((loc, ref), (bad, None)):
                dloc = torch.from_numpy(l).to(DEV)
                outs = []
                for key, lds in ((0, 1), (1, 1), (1, 0), (0, 0)):
                    NW.tune_set(37, key)
                    NW.tune_set(17, lds)
                    out = torch.full((B, N, H, W), float("nan"), device=DEV)
                    NW.ScatterConnectionForward([dx, dloc], [out], typ)
                    outs.append(out.cpu().numpy())
                for o in outs[1:]:
                    assert np.array_equal(outs[0], o), typ
                if want is not None:
                    assert np.array_equal(outs[1], want), typ
                else:
                    assert not np.isnan(outs[1]).any()
    finally:
        NW.tune_set(37, 1)
        NW.tune_set(17, 1)


@pytest.mark.parametrize("B,M,N,H,W", [(16, 256, 64, 64, 64), (24, 100, 16, 64, 64), (8, 37, 12, 64, 32), (40, 64, 64, 32, 32), (3, 50, 8, 64, 64)])
def test_scatter_backward_xcd_order_bit_exact(B, M, N, H, W):
    """Tune key 38 (round 4): the channel groups of a batch element take consecutive slots of ONE XCD (logical index
    (i % 8) * (total / 8) + i / 8) so that their pieces of the same grad_x lines meet in one L2.  A pure re-labelling of which
    workgroup does what: launch order (0), the default rule (1) and always (2) give the same bits, also where the grid is no
    multiple of 8 (no re-labelling) and with out-of-range locations (zero gradient)."""
    import cabi as C
    import hpc_torch_utils_network as NW
    g = torch.Generator(device=DEV).manual_seed(B * 131 + M)
    go = torch.randn(B, N, H, W, device=DEV, generator=g)
    loc = torch.stack([torch.randint(-1, H + 1, (B, M), device=DEV, generator=g), torch.randint(0, W, (B, M), device=DEV, generator=g)], -1)
    s = torch.cuda.current_stream().cuda_stream
    outs = []
    try:
        for key in (0, 1, 2):
            NW.tune_set(38, key)
            gx = torch.full((B, M, N), float("nan"), device=DEV)
            assert C.lib.hpc_rll_scatter_connection_backward(go.data_ptr(), loc.data_ptr(), gx.data_ptr(), B, M, N, H, W, s) == 0
            outs.append(gx.cpu())
    finally:
        NW.tune_set(38, 1)
    ok = (loc[..., 0] >= 0) & (loc[..., 0] < H)
    want = go.permute(0, 2, 3, 1)[torch.arange(B, device=DEV)[:, None], loc[..., 0].clamp(0, H - 1), loc[..., 1]] * ok[..., None]
    assert torch.equal(outs[0], want.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("B,M,N,H,W", [(16, 256, 64, 64, 64), (24, 1024, 128, 64, 64), (9, 37, 12, 64, 32), (40, 64, 64, 32, 32), (3, 4096, 8, 64, 64),
                                       (5, 200, 24, 40, 52), (7, 300, 32, 128, 128), (8, 77, 20, 3, 4), (2, 50, 256, 16, 64), (11, 1, 4, 64, 32)])
def test_scatter_backward_spatial_tile_kernel_bit_exact(B, M, N, H, W):
    """Round 6, tune key 40: the backward by SPATIAL tiles (a workgroup stages TR map rows of all N planes and writes the whole
    rows of the entities that fall into them) against the plane kernel (key 40 = 1) and the gather computed by torch: the same
    bits.  Shapes: the rule's own (64 x 64 maps, M * 16 >= H * W), two rows per tile (N = 128), maps whose row count is no
    multiple of the tile (H = 40, 3), tiles that are no power of two cells (W = 52: no swizzle), one row per tile (N = 256),
    M = 4096 (four locations per thread), M = 1, out-of-range locations in both coordinates (zero rows, dealt over the tiles),
    shapes where the rule keeps the plane kernel (forced with key 40 = 2), B a multiple of 8 (the plane kernel's XCD order 2) and not."""
    import cabi as C
    import hpc_torch_utils_network as NW
    g = torch.Generator(device=DEV).manual_seed(B * 977 + M)
    go = torch.randn(B, N, H, W, device=DEV, generator=g)
    loc = torch.stack([torch.randint(-1, H + 1, (B, M), device=DEV, generator=g), torch.randint(-1, W + 1, (B, M), device=DEV, generator=g)], -1)
    s = torch.cuda.current_stream().cuda_stream
    outs = {}
    try:
        for key in (1, 2, 0):
            NW.tune_set(40, key)
            gx = torch.full((B, M, N), float("nan"), device=DEV)
            assert C.lib.hpc_rll_scatter_connection_backward(go.data_ptr(), loc.data_ptr(), gx.data_ptr(), B, M, N, H, W, s) == 0
            outs[key] = gx.cpu()
    finally:
        NW.tune_set(40, 0)
    ok = (loc[..., 0] >= 0) & (loc[..., 0] < H) & (loc[..., 1] >= 0) & (loc[..., 1] < W)
    want = go.permute(0, 2, 3, 1)[torch.arange(B, device=DEV)[:, None], loc[..., 0].clamp(0, H - 1), loc[..., 1].clamp(0, W - 1)] * ok[..., None]
    assert torch.equal(outs[1], want.cpu())
    assert torch.equal(outs[2], outs[1]) and torch.equal(outs[0], outs[1])


def test_scatter_backward_of_an_empty_map_is_zero():
    """C ABI edge (round 6): H * W == 0 -- every location is out of range, the gradient rows are zeros (until round 5 the launcher
    divided by the plane size)."""
    import cabi as C
    B, M, N = 3, 5, 8
    loc = torch.zeros(B, M, 2, dtype=torch.int64, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    for H, W in ((0, 4), (4, 0), (0, 0)):
        gx = torch.full((B, M, N), float("nan"), device=DEV)
        assert C.lib.hpc_rll_scatter_connection_backward(0, loc.data_ptr(), gx.data_ptr(), B, M, N, H, W, s) == 0
        assert float(gx.abs().max()) == 0.0
