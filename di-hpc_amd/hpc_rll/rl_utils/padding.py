"""``hpc_rll.rl_utils.padding`` -- drop-in for /root/reference/hpc_rll/rl_utils/padding.py:14-180:
``Padding{1,2,3}D(inputs, mode='constant', value=0, group=1, group_mode='sample')`` and
``UnPadding{1,2,3}D(x, shapes)`` with the reference's return conventions: ``(new_x, mask, shapes)`` where ``mask`` is
int32 and ``shapes`` is a FLAT int list (rank ints per tensor), or ``[tuple(new_x), tuple(mask), tuple(shapes)]`` when
``group > 1`` (inputs are then sorted by element count first)."""
from typing import List, Union

import torch

import hpc_rl_utils


def _padding(inputs, mode, value, group, group_mode, rank):
    assert mode in ['constant'], mode
    assert group_mode in ['sample', 'oracle'], group_mode
    assert group >= 1, group
    if group > 1:   # ordering by element count, the split policy and the per-group launches: one native call
        return hpc_rl_utils.padding_grouped(inputs, value, group, group_mode, rank)
    pad = {1: hpc_rl_utils.Pad1DForward, 2: hpc_rl_utils.Pad2DForward, 3: hpc_rl_utils.Pad3DForward}[rank]
    if rank == 1:
        shapes = list(map(torch.Tensor.numel, inputs))        # C-level loop, ~0.12 us per tensor (t.shape: 0.9 us)
    else:
        shapes = [int(v) for t in inputs for v in t.shape]
    new_x, mask = pad(inputs, value)
    return new_x, mask, shapes


def _unpadding(x, shapes, rank):
    un = {1: hpc_rl_utils.Unpad1DForward, 2: hpc_rl_utils.Unpad2DForward, 3: hpc_rl_utils.Unpad3DForward}[rank]
    if isinstance(x, torch.Tensor):
        return un(x, shapes)
    ret = []
    for t, s in zip(x, shapes):
        ret.append(un(t, s))
    return sum(ret, [])


def Padding1D(inputs: List[torch.Tensor], mode='constant', value: int = 0, group: int = 1, group_mode='sample'):
    return _padding(inputs, mode, value, group, group_mode, 1)


def UnPadding1D(x: Union[torch.Tensor, List[torch.Tensor]], shapes: Union[List, List[List]]) -> List[torch.Tensor]:
    return _unpadding(x, shapes, 1)


def Padding2D(inputs: List[torch.Tensor], mode='constant', value: int = 0, group: int = 1, group_mode='sample'):
    return _padding(inputs, mode, value, group, group_mode, 2)


def UnPadding2D(x: Union[torch.Tensor, List[torch.Tensor]], shapes: Union[List, List[List]]) -> List[torch.Tensor]:
    return _unpadding(x, shapes, 2)


def Padding3D(inputs: List[torch.Tensor], mode='constant', value: int = 0, group: int = 1, group_mode='sample'):
    return _padding(inputs, mode, value, group, group_mode, 3)


def UnPadding3D(x: Union[torch.Tensor, List[torch.Tensor]], shapes: Union[List, List[List]]) -> List[torch.Tensor]:
    return _unpadding(x, shapes, 3)


# ---------------------------------------------------------------------------------------------------------------------
# Packed (CSR-style) variants -- not in the reference.  A python list of a million tiny tensors costs ~2 us of host time
# per tensor just to be described to the kernel; at BASELINE.json's configs[4] scale (~1M entities per batch) the natural
# device layout is ONE flat buffer plus a device vector of lengths.  Same kernels, table built on the device, no host
# loop and (when ``max_len`` is given) no host synchronisation.
# ---------------------------------------------------------------------------------------------------------------------
def _validate_packed(numel: int, lengths: torch.Tensor, max_len, what: str):
    """The packed entry points trust ``lengths`` / ``max_len`` / ``total`` (no host synchronisation); ``validate=True``
    checks them with one: every length in [0, max_len] and sum(lengths) == the flat element count."""
    if lengths.numel() == 0:
        return
    lo, hi, tot = int(lengths.min()), int(lengths.max()), int(lengths.sum())
    if lo < 0 or (max_len is not None and hi > max_len) or (numel is not None and tot != numel):
        raise ValueError(f"{what}: lengths in [{lo}, {hi}] (max_len {max_len}), sum {tot} (flat elements {numel})")


def Padding1DPacked(flat: torch.Tensor, lengths: torch.Tensor, max_len: int = None, value: int = 0, group: int = 1,
                    group_mode: str = 'oracle', seed: int = None, validate: bool = False):
    """flat (sum(lengths),) fp32, lengths (n,) int64 on the same GPU -> (new_x (n,max_len) fp32, mask (n,max_len) int32).
    Row i of new_x holds flat[offset_i : offset_i + lengths[i]] followed by ``value``.  The offsets are an exclusive scan
    done on the device inside the extension; ``max_len=None`` costs the only host sync (lengths.max()).

    ``group > 1`` (the reference's bucketing, hpc_rll/rl_utils/padding.py:20-45, entirely on the device): the rows are
    taken in sorted order (ascending length, original order among equal lengths -- python's ``sorted``), split into at
    most ``group`` buckets by ``group_mode`` ('oracle': the padded-element-minimising DP of
    hpc_rll/origin/padding.py:11-50 with its tie rule; 'sample': random cuts) and every bucket is padded to its own
    width by ONE launch.  Returns ``[tuple(new_x_g), tuple(mask_g), tuple(lengths_g), order]``: bucket g holds the
    original rows ``order[cut_g : cut_{g+1}]``, ``lengths_g`` their lengths.  ``max_len`` <= 16384; one host sync (the
    bucket shapes).

    Preconditions (unchecked unless ``validate=True``, which costs a host sync): 0 <= lengths[i] <= max_len and
    sum(lengths) == flat.numel(); a longer row is truncated to ``max_len`` columns, a sum beyond ``flat`` reads past it."""
    if validate:
        _validate_packed(flat.numel(), lengths, max_len, "Padding1DPacked")
    if group > 1:
        xs, ms, ls, (order,) = hpc_rl_utils.pad1d_packed_grouped(flat, lengths, max_len, value, group, group_mode, seed)
        return [tuple(xs), tuple(ms), tuple(ls), order]
    new_x, mask = hpc_rl_utils.pad1d_packed(flat, lengths, max_len, value)
    return new_x, mask


def UnPadding1DPacked(x: torch.Tensor, lengths: torch.Tensor, total: int = None, validate: bool = False) -> torch.Tensor:
    """Inverse of :func:`Padding1DPacked`: x (n,max_len), lengths (n,) int64 -> flat (sum(lengths),).  ``total`` (the sum
    of the lengths) saves the host sync; elements of a too large ``total`` are left unwritten, lengths beyond
    ``x.shape[1]`` are not read (``validate=True`` raises instead)."""
    if validate:
        _validate_packed(total, lengths, x.shape[1], "UnPadding1DPacked")
    return hpc_rl_utils.unpad1d_packed(x, lengths, total)


def UnPadding1DPackedGrouped(xs, lengths_groups, order: torch.Tensor, total: int = None) -> torch.Tensor:
    """Inverse of ``Padding1DPacked(..., group > 1)``: the buckets ``xs[g]`` (cnt_g, width_g), their row lengths
    ``lengths_groups[g]`` (cnt_g,) and ``order`` (n,) as returned by it -> the flat values in the ORIGINAL row order.
    The buckets are laid side by side as (n, max width) sorted rows, the rows are put back in their original order with
    one ``index_select`` through the inverse permutation, and the packed unpad kernel does the rest."""
    n = order.numel()
    width = max([int(x.shape[1]) for x in xs] + [0])
    rows = xs[0].new_zeros((n, width)) if len(xs) else order.new_zeros((0, 0), dtype=torch.float32)
    k = 0
    for x in xs:
        rows[k:k + x.shape[0], :x.shape[1]] = x
        k += x.shape[0]
    inv = torch.empty_like(order)
    inv[order] = torch.arange(n, device=order.device, dtype=order.dtype)
    lengths_sorted = torch.cat(list(lengths_groups)) if len(lengths_groups) else order.new_zeros((0,))
    return hpc_rl_utils.unpad1d_packed(rows.index_select(0, inv), lengths_sorted.index_select(0, inv), total)
