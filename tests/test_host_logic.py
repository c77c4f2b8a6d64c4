"""CPU tests of the host-side logic that lives in the C-ABI library (no GPU needed): the group-split policies of
Pad*D (reference: src/rl_utils/padding.cu:8-108) against the oracle DP and the golden fixtures."""
import ctypes
import random

import numpy as np
import pytest
import torch

from oracle import ref_torch as R


def _split(nl, group):
    import hpc_rl_utils as U
    return U.oracle_split_group([torch.empty(n) for n in nl], group)


def test_oracle_split_matches_golden(golden):
    g = golden("padding")
    for j in range(int(g["split_n"])):
        nl = [int(v) for v in g[f"split{j}_numels"]]
        res = _split(nl, int(g[f"split{j}_group"]))
        assert res[-1] == [int(v) for v in g[f"split{j}_pos"]]
        # group shapes are the per-group maxima
        for k, shape in enumerate(res[:-1]):
            assert shape == [max(nl[res[-1][k]:res[-1][k + 1]])]


@pytest.mark.parametrize("n,group", [(600, 4), (1500, 7), (2000, 3), (513, 16), (700, 1)])
def test_fast_split_equals_quadratic_dp(n, group):
    """n > 512 sorted 1-D lists take the O(n log n) divide-and-conquer layer; it must return the quadratic DP's answer,
    tie-breaking (smallest split point) included.  Lengths are drawn from a small range to force many ties."""
    rng = np.random.default_rng(n)
    nl = sorted(int(v) for v in rng.integers(1, 40, n))
    assert _split(nl, group)[-1] == R.oracle_split_group(nl, group)


def test_split_multi_dim_uses_elementwise_max():
    import hpc_rl_utils as U
    xs = [torch.empty(2, 9), torch.empty(5, 4), torch.empty(3, 8), torch.empty(6, 6)]
    xs = sorted(xs, key=lambda t: t.numel())
    res = U.oracle_split_group(xs, 2)
    pos = res[-1]
    for k, shape in enumerate(res[:-1]):
        grp = xs[pos[k]:pos[k + 1]]
        assert shape == [max(t.shape[0] for t in grp), max(t.shape[1] for t in grp)]


def test_sample_split_is_valid_and_seeded():
    import hpc_rl_utils as U
    xs = [torch.empty(n) for n in sorted([5, 9, 9, 12, 30, 31, 40, 100, 100, 101])]
    random.seed(3)
    a = U.sample_split_group(xs, 4)
    random.seed(3)
    b = U.sample_split_group(xs, 4)
    assert a == b
    pos = a[-1]
    assert pos[0] == 0 and pos[-1] == len(xs) and all(p < q for p, q in zip(pos, pos[1:]))
    for k, shape in enumerate(a[:-1]):
        assert shape == [max(t.numel() for t in xs[pos[k]:pos[k + 1]])]
    assert U.sample_split_group(xs[:2], 3)[-1] == [0, 2]          # n = 2: the reference divides by zero here


def _reference_sample_split(numels, group, seed):
    """src/rl_utils/padding.cu:8-43 restated in python for sorted 1-D lists, with the library's splitmix64 stream in place of
    C rand(): group-1 cuts in [1, N-2] (a cut equal to the PREVIOUS draw is re-drawn), sorted, N-1 appended; a cut whose
    group has the same shape as the previous accepted group is skipped WITHOUT advancing last_idx.  (A repeated cut
    position -- where the reference emits an empty group of shape -1 -- is dropped, as the library documents.)"""
    mask = (1 << 64) - 1
    n = len(numels)
    cuts = []
    if n >= 3:
        last = -1
        for _ in range(group - 1):
            now = last
            if n == 3:
                now = 1
            else:
                while now == last:
                    seed = (seed + 0x9E3779B97F4A7C15) & mask
                    z = seed
                    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
                    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
                    z ^= z >> 31
                    now = z % (n - 2) + 1
            cuts.append(now)
            last = now
        cuts.sort()
    cuts.append(n - 1)
    shapes, idxs, last_idx = [], [], -1
    for idx in cuts:
        if idx <= last_idx:
            continue
        shape = max(numels[last_idx + 1:idx + 1])
        if shapes and shape == shapes[-1]:
            continue
        shapes.append(shape)
        idxs.append(last_idx + 1)
        last_idx = idx
    idxs.append(n)
    return shapes, idxs


@pytest.mark.parametrize("n,group,seed", [(10, 4, 3), (200, 8, 11), (50, 16, 5), (3, 4, 1), (1000, 63, 7), (40, 6, 123456789)])
def test_sample_split_mirrors_the_reference_rule(n, group, seed):
    """same cuts -> same boundaries as the reference's loop, including its equal-shape skip (rows of a skipped
    cut join the NEXT group).  Lengths from a small range, so that equal-shape neighbours are common."""
    import hpc_rl_utils as U
    rng = np.random.default_rng(n + group)
    nl = sorted(int(v) for v in rng.integers(1, 9, n))
    res = U.sample_split_group([torch.empty(v) for v in nl], group, seed)
    shapes, idxs = _reference_sample_split(nl, group, seed)
    assert res[-1] == idxs and [r[0] for r in res[:-1]] == shapes


def test_large_list_is_fast():
    import time
    rng = np.random.default_rng(1)
    nl = sorted(int(v) for v in rng.integers(32, 128, 200000))
    t = time.time()
    pos = _split(nl, 8)[-1]
    assert time.time() - t < 20 and pos[0] == 0 and pos[-1] == len(nl) and len(pos) == 9


def test_pad_and_unpad_host_tables():
    """The (pointer, shape) tables the pad / unpad kernels consume are built by C++ loops in the extension
    (hpc_rl_utils._pad_table / _unpad_table expose them); pure host logic, checked here against the obvious
    per-tensor loops."""
    import hpc_rl_utils as U
    rng = np.random.default_rng(0)
    for rank in (1, 2, 3):
        shapes = [tuple(int(v) for v in rng.integers(1, 6, rank)) for _ in range(17)]
        xs = [torch.zeros(*s) for s in shapes]
        table, sh = (t.numpy() for t in U._pad_table(xs, rank))
        assert table.shape == (17, 4) and sh.shape == (17, rank)
        for i, (t, s_) in enumerate(zip(xs, shapes)):
            assert table[i, 0] == t.data_ptr()
            assert tuple(table[i, 1:]) == (1,) * (3 - rank) + s_
            assert tuple(sh[i]) == s_
        flat_shapes = [d for s_ in shapes for d in s_]
        lim = tuple(int(v) for v in np.max(np.array(shapes), axis=0))
        tab, numel, offs, sh2 = (t.numpy() for t in U._unpad_table(flat_shapes, lim, rank))
        want = [int(np.prod(s_)) for s_ in shapes]
        assert numel.tolist() == want and offs.tolist() == [0] + list(np.cumsum(want))
        assert tab[:, 0].tolist() == offs[:-1].tolist()
        assert [tuple(r[4 - rank:]) for r in tab] == shapes and np.all(tab[:, 1:4 - rank] == 1)
        bad = list(flat_shapes)
        bad[0] = lim[0] + 1                                      # does not fit the padded tensor
        with pytest.raises(RuntimeError):
            U._unpad_table(bad, lim, rank)
    tab, numel, offs, _ = (t.numpy() for t in U._unpad_table([], (4,), 1))   # empty list
    assert tab.shape == (0, 4) and offs.tolist() == [0]


@pytest.mark.parametrize("rank,n,group", [(2, 700, 5), (3, 1200, 4), (2, 100000, 8)])
def test_long_multi_dim_lists_split_in_n_log_n(rank, n, group):
    """rank-2/3 lists beyond 512 tensors no longer need the O(group * n^2) table.  They take
    the monotone divide-and-conquer DP on the element count, i.e. the cost model of the reference's OWN python oracle
    (hpc_rll/origin/padding.py:16-17: arr[end] * count over the numel-sorted list): the positions must equal that
    oracle's, the group shapes are the true per-dimension maxima, and 100k tensors finish in seconds."""
    import time
    import hpc_rl_utils as U
    rng = np.random.default_rng(rank * 1000 + n)
    shapes = [tuple(int(v) for v in rng.integers(1, 12, rank)) for _ in range(n)]
    shapes.sort(key=lambda s: int(np.prod(s)))
    xs = [torch.empty(s) for s in shapes]
    t0 = time.time()
    res = U.oracle_split_group(xs, group)
    assert time.time() - t0 < 30
    pos = res[-1]
    assert pos[0] == 0 and pos[-1] == n and all(p < q for p, q in zip(pos, pos[1:]))
    for k, shape in enumerate(res[:-1]):
        grp = shapes[pos[k]:pos[k + 1]]
        assert shape == [max(s[d] for s in grp) for d in range(rank)]
    if n <= 2000:      # the python restatement of origin.oracle_split_group on the numels (quadratic: small n only)
        assert pos == R.oracle_split_group([int(np.prod(s)) for s in shapes], group)


def test_unsorted_long_list_is_refused_not_quadratic():
    import hpc_rl_utils as U
    xs = [torch.empty(3, (i * 7919) % 50 + 1) for i in range(20001)]      # not sorted by numel
    with pytest.raises(RuntimeError):
        U.oracle_split_group(xs, 4)


def _split_pos_np(nl, group, algo):
    """positions from the C ABI directly (no tensor list): algo 0 = runs of equal keys, 1 = the element-level paths"""
    import cabi as N
    n = len(nl)
    sizes = (ctypes.c_int32 * n)(*nl)
    shapes = (ctypes.c_int32 * max(group, 1))()
    pos = (ctypes.c_int32 * (group + 1))()
    assert N.lib.hpc_rll_tune_set(22, algo) == 0
    try:
        ng = N.lib.hpc_rll_oracle_split_group(sizes, n, 1, group, shapes, pos)
    finally:
        N.lib.hpc_rll_tune_set(22, 0)
    assert ng >= 1, ng
    return list(pos[:ng + 1]), list(shapes[:ng])


def test_run_level_split_equals_reference_dp_including_forced_cuts():
    """Round 3 (f-3): the split works on RUNS of equal keys (csrc/pad_group.hpp).  Small lists against the python
    restatement of hpc_rll.origin.padding.oracle_split_group (the quadratic element DP, smallest split point on ties):
    many ties, fewer distinct lengths than groups (cuts forced inside runs), single-run lists, group > n."""
    rng = np.random.default_rng(7)
    cases = [([5] * 9, 4), ([5] * 3, 8), ([1, 1, 1, 2, 2, 9], 5), ([3, 3, 3, 3, 7, 7, 7, 7, 7, 7], 4), ([2], 3), ([4, 4], 2),
             ([1, 2, 3, 4, 5, 6, 7, 8], 8), ([1, 2, 3, 4, 5, 6, 7, 8], 3), ([0, 0, 1, 1, 1, 5], 4)]
    for _ in range(300):
        n = int(rng.integers(1, 60))
        hi = int(rng.integers(1, 12))
        cases.append((sorted(int(v) for v in rng.integers(0 if hi > 3 else 1, hi + 1, n)), int(rng.integers(1, 12))))
    for nl, group in cases:
        want = R.oracle_split_group(nl, min(group, len(nl)))
        pos, shapes = _split_pos_np(nl, group, 0)
        assert pos == want, (nl, group, pos, want)
        assert shapes == [max(nl[a:b]) for a, b in zip(pos, pos[1:])]
        if len(nl) > 0:
            assert _split_pos_np(nl, group, 1)[0] == want, (nl, group)


@pytest.mark.parametrize("n,hi,group", [(5000, 6, 8), (20000, 40, 8), (200000, 128, 8), (200000, 3, 16), (50000, 5000, 12),
                                        (30000, 1, 5)])
def test_run_level_split_equals_element_level_dp_on_long_lists(n, hi, group):
    """... and long lists (up to 2 x 10^5) against the round-2 divide-and-conquer DP over ELEMENTS (tune key 22 = 1)."""
    rng = np.random.default_rng(n + hi)
    nl = sorted(int(v) for v in rng.integers(1, hi + 1, n))
    a = _split_pos_np(nl, group, 0)
    b = _split_pos_np(nl, group, 1)
    assert a == b


def test_lstm_loads_a_reference_checkpoint_and_keeps_its_own_small():
    """the reference LSTM registers ~17 scratch buffers (rnn.py:117-141) that land in its state_dict; a
    checkpoint written by it must load (strict) into this module, which itself saves the five parameters only."""
    from hpc_rll.torch_utils.network.rnn import LSTM, _REFERENCE_SCRATCH
    m = LSTM(4, 2, 3, 5, 2)
    sd = m.state_dict()
    assert sorted(sd) == ["bias", "ln_beta", "ln_gamma", "wh", "wx"]
    orig = {k: v.clone() for k, v in sd.items()}
    ref_sd = {k: v + 1.0 for k, v in orig.items()}
    for name in _REFERENCE_SCRATCH:
        ref_sd[name] = torch.zeros(3)
    m.load_state_dict(ref_sd, strict=True)
    assert torch.equal(m.wx, orig["wx"] + 1.0)
    net = torch.nn.Sequential(torch.nn.Linear(2, 2), LSTM(4, 2, 3, 5, 1))            # nested prefix
    nsd = net.state_dict()
    nsd["1.xbuf"] = torch.zeros(1)
    net.load_state_dict(nsd, strict=True)
