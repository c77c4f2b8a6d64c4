"""``hpc_torch_utils_network`` -- the reference's second native extension module
(src/torch_utils/network/entry.cpp:8-13), as a thin binding over the C ABI of libhpc_rll_hip.so.
Same ``Fn(inputs, outputs, scalars...)`` convention as the reference."""
import torch

from hpc_rll import _native as N

_lib = N.lib
F32, I64 = torch.float32, torch.int64


def ScatterConnectionForward(inputs, outputs, scatter_type: str) -> None:
    """inputs = [x (B,M,N) fp32, location (B,M,2) int64 (y,x)], outputs = [out (B,N,H,W)].
    Reference: src/torch_utils/network/scatter_connection.cu:8-49.  ``out`` is fully overwritten."""
    x, location = inputs
    (out,) = outputs
    N.require(x, "x")
    if x.dim() != 3:
        raise RuntimeError(f"x: expected (B,M,N), got {tuple(x.shape)}")
    B, M, NA = x.shape
    dev = x.device
    N.require(location, "location", dtype=I64, shape=(B, M, 2), device=dev)
    N.require(out, "output", device=dev)
    if out.dim() != 4 or out.shape[0] != B or out.shape[1] != NA:
        raise RuntimeError(f"output: expected ({B},{NA},H,W), got {tuple(out.shape)}")
    H, W = out.shape[2], out.shape[3]
    if scatter_type not in ("cover", "add"):
        raise RuntimeError(f"scatter_type: {scatter_type!r}")
    ws = torch.empty(_lib.hpc_rll_scatter_workspace_ints(B, M, H, W), dtype=torch.int32, device=dev)
    N.call("hpc_rll_scatter_connection_forward", dev, x.data_ptr(), location.data_ptr(), out.data_ptr(),
           ws.data_ptr(), B, M, NA, H, W, 1 if scatter_type == "add" else 0)


def ScatterConnectionBackward(inputs, outputs) -> None:
    """inputs = [grad_out (B,N,H,W), location (B,M,2)], outputs = [grad_x (B,M,N)].
    Reference: src/torch_utils/network/scatter_connection.cu:51-73."""
    grad_out, location = inputs
    (grad_x,) = outputs
    N.require(grad_out, "grad_out")
    B, NA, H, W = grad_out.shape
    dev = grad_out.device
    M = location.shape[1]
    N.require(location, "location", dtype=I64, shape=(B, M, 2), device=dev)
    N.require(grad_x, "grad_x", shape=(B, M, NA), device=dev)
    N.call("hpc_rll_scatter_connection_backward", dev, grad_out.data_ptr(), location.data_ptr(), grad_x.data_ptr(),
           B, M, NA, H, W)
