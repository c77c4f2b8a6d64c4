"""``hpc_torch_utils_network`` -- the reference's second native extension module
(src/torch_utils/network/entry.cpp:8-13), as a thin binding over the C ABI of libhpc_rll_hip.so.
Same ``Fn(inputs, outputs, scalars...)`` convention as the reference."""
import torch

from hpc_rll import _native as N

_lib = N.lib
F32, I64 = torch.float32, torch.int64


def lstm_workspace(S, B, I, H, L, dropout, dev):
    n = _lib.hpc_rll_lstm_workspace_floats(int(S), int(B), int(I), int(H), int(L), float(dropout))
    if n < 0:
        raise RuntimeError("lstm_workspace: invalid sizes")
    return torch.empty(n, dtype=F32, device=dev)


def _lstm_dims(x, h0, wx, wh):
    S, B, I = x.shape
    L, _, H = h0.shape
    G = 4 * H
    if wx.numel() != (I + (L - 1) * H) * G or wh.numel() != L * H * G:
        raise RuntimeError(f"wx/wh: {wx.numel()}/{wh.numel()} elements do not match I={I} H={H} L={L}")
    return S, B, I, H, L


def LstmForward(inputs, outputs, dropout: float, seed: int = 0) -> None:
    """inputs = [x (S,B,I), h0 (L,B,H), c0 (L,B,H), wx (flat), wh (flat), bias (L*4H), ln_gamma (L,8H), ln_beta (L,8H)]
    (the reference's input order, torch_utils/network/rnn.py:20); outputs = [y (S,B,H), hn (L,B,H), cn (L,B,H),
    ws = lstm_workspace(...)] -- the reference's ten scratch buffers (xbuf, hbuf, hn, cn, ifog, ym, ln_in, ln_mean,
    ln_rstd, dropout_mask) live in ``ws``.  Reference: src/torch_utils/network/lstm.cu:29-186."""
    x, h0, c0, wx, wh, bias, gamma, beta = inputs
    y, hn, cn, ws = outputs
    N.require(x, "x")
    dev = x.device
    N.require(h0, "h0", device=dev)
    S, B, I, H, L = _lstm_dims(x, h0, wx, wh)
    G = 4 * H
    N.require(c0, "c0", shape=(L, B, H), device=dev)
    for t, nm in ((wx, "wx"), (wh, "wh"), (bias, "bias"), (gamma, "ln_gamma"), (beta, "ln_beta")):
        N.require(t, nm, device=dev)
    if bias.numel() != L * G or gamma.numel() != L * 2 * G or beta.numel() != L * 2 * G:
        raise RuntimeError("bias / ln_gamma / ln_beta: wrong number of elements")
    N.require(y, "y", shape=(S, B, H), device=dev)
    N.require(hn, "hn", shape=(L, B, H), device=dev)
    N.require(cn, "cn", shape=(L, B, H), device=dev)
    N.require(ws, "ws", shape=(_lib.hpc_rll_lstm_workspace_floats(S, B, I, H, L, float(dropout)),), device=dev)
    N.call("hpc_rll_lstm_forward", dev, x.data_ptr(), h0.data_ptr(), c0.data_ptr(), wx.data_ptr(), wh.data_ptr(),
           bias.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), hn.data_ptr(), cn.data_ptr(),
           ws.data_ptr(), S, B, I, H, L, float(dropout), int(seed))


def LstmBackward(inputs, outputs, dropout: float, seed: int = 0) -> None:
    """inputs = [dy (S,B,H)|None, dhn (L,B,H)|None, dcn (L,B,H)|None, x, h0, c0, wx, wh, ln_gamma, ws];
    outputs = [dx|None, dh0, dc0, dwx, dwh, dbias, d_ln_gamma, d_ln_beta].  Reference: lstm.cu:188-379 (which zeroes the
    incoming dhn/dcn; here they are honoured).  dx = None skips the input-gradient product of layer 0 (x needs no grad)."""
    dy, dhn, dcn, x, h0, c0, wx, wh, gamma, ws = inputs
    dx, dh0, dc0, dwx, dwh, dbias, dgamma, dbeta = outputs
    dev = x.device
    S, B, I, H, L = _lstm_dims(x, h0, wx, wh)
    if dy is not None:
        N.require(dy, "dy", shape=(S, B, H), device=dev)
    if dhn is not None:
        N.require(dhn, "dhn", shape=(L, B, H), device=dev)
    if dcn is not None:
        N.require(dcn, "dcn", shape=(L, B, H), device=dev)
    if dx is not None:
        N.require(dx, "dx", shape=(S, B, I), device=dev)
    N.require(dh0, "dh0", shape=(L, B, H), device=dev)
    N.require(dc0, "dc0", shape=(L, B, H), device=dev)
    for t, ref, nm in ((dwx, wx, "dwx"), (dwh, wh, "dwh"), (dgamma, gamma, "d_ln_gamma"), (dbeta, gamma, "d_ln_beta")):
        N.require(t, nm, device=dev)
        if t.numel() != ref.numel():
            raise RuntimeError(f"{nm}: {t.numel()} elements, expected {ref.numel()}")
    N.require(dbias, "dbias", device=dev)
    N.call("hpc_rll_lstm_backward", dev, N.ptr(dy), N.ptr(dhn), N.ptr(dcn), x.data_ptr(), h0.data_ptr(), c0.data_ptr(),
           wx.data_ptr(), wh.data_ptr(), gamma.data_ptr(), ws.data_ptr(), N.ptr(dx), dh0.data_ptr(), dc0.data_ptr(),
           dwx.data_ptr(), dwh.data_ptr(), dbias.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), S, B, I, H, L,
           float(dropout), int(seed))


def gemm_f32(a: torch.Tensor, b: torch.Tensor, out=None, accumulate=False) -> torch.Tensor:
    """out (M,N) (+)= a (M,K) @ b (K,N) in exact fp32 on the matrix cores; a and b may be arbitrary 2-D strided views
    (e.g. ``w.t()``), which is how the NN / NT / TN layouts of the LSTM are expressed."""
    M, K = a.shape
    K2, Nn = b.shape
    assert K == K2 and a.is_cuda and b.is_cuda and a.dtype == F32 and b.dtype == F32
    if out is None:
        out = torch.empty(M, Nn, dtype=F32, device=a.device)
    N.require(out, "out", shape=(M, Nn), device=a.device)
    N.call("hpc_rll_gemm_f32", a.device, a.data_ptr(), b.data_ptr(), out.data_ptr(), M, Nn, K, a.stride(0), a.stride(1),
           b.stride(0), b.stride(1), out.stride(0), int(bool(accumulate)))
    return out


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
