"""``hpc_models`` -- the reference's third native extension (src/models/entry.cpp:8-12): three forward-only AlphaStar
actor-critic inference helpers, as a thin binding over the C ABI.  Same ``Fn(inputs, outputs)`` convention and tensor
order as the reference (src/models/actor_critic.cu:8-83)."""
import torch

from hpc_rll import _native as N

F32, I64 = torch.float32, torch.int64


def actor_critic_update_ae(inputs, outputs) -> None:
    """inputs = [key_embeddings (B,E,D), sample_entity (B,) int64, entity_num (B,) int64];
    outputs = [autoregressive_embedding (B,D)] which is updated IN PLACE:
    ae[b] += key_embeddings[b, sample_entity[b]] unless sample_entity[b] == entity_num[b] (the "end" action)."""
    key, sample, num = inputs
    (ae,) = outputs
    N.require(key, "key_embeddings")
    B, E, D = key.shape
    dev = key.device
    N.require(sample, "sample_entity", dtype=I64, shape=(B,), device=dev)
    N.require(num, "entity_num", dtype=I64, shape=(B,), device=dev)
    N.require(ae, "autoregressive_embedding", shape=(B, D), device=dev)
    N.call("hpc_rll_actor_critic_update_ae", dev, key.data_ptr(), sample.data_ptr(), num.data_ptr(), ae.data_ptr(), B, E, D)


def actor_critic_lstm_activation(inputs, outputs) -> None:
    """inputs = [lstm_ih (B,4H), lstm_hh (B,4H), bias (4H,)], outputs = [h (B,H) written, c (B,H) updated in place];
    gate order i,f,g,o (torch.nn.LSTM)."""
    ih, hh, bias = inputs
    h, c = outputs
    N.require(ih, "lstm_ih")
    B, G = ih.shape
    H = G // 4
    dev = ih.device
    N.require(hh, "lstm_hh", shape=(B, G), device=dev)
    N.require(bias, "lstm_bias", device=dev)
    if bias.numel() != G:
        raise RuntimeError(f"lstm_bias: {bias.numel()} elements, expected {G}")
    N.require(h, "lstm_hx", device=dev)
    N.require(c, "lstm_cx", device=dev)
    if h.numel() != B * H or c.numel() != B * H:
        raise RuntimeError("lstm_hx / lstm_cx: expected B*H elements")
    N.call("hpc_rll_actor_critic_lstm_activation", dev, ih.data_ptr(), hh.data_ptr(), bias.data_ptr(), h.data_ptr(),
           c.data_ptr(), B, H)


def actor_critic_pre_sample(inputs, outputs) -> None:
    """inputs = [mat (B,E,H), vec (.., B, H) (any leading singleton dims), mask (B,E) bool]; outputs = [out (B,E)]:
    out = where(mask, (mat * vec[:,None,:]).sum(-1), -1e9) / 0.8."""
    mat, vec, mask = inputs
    (out,) = outputs
    N.require(mat, "mat")
    B, E, H = mat.shape
    dev = mat.device
    N.require(vec, "vec", device=dev)
    if vec.numel() != B * H:
        raise RuntimeError(f"vec: {tuple(vec.shape)} does not hold B*H = {B * H} values")
    N.require(mask, "mask", dtype=torch.bool, shape=(B, E), device=dev)
    N.require(out, "output", shape=(B, E), device=dev)
    N.call("hpc_rll_actor_critic_pre_sample", dev, mat.data_ptr(), vec.data_ptr(), mask.data_ptr(), out.data_ptr(), B, E, H,
           -1e9, 0.8)
