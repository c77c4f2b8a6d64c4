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
