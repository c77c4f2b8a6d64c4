// models.cpp -- the compiled `hpc_models` extension module (reference: src/models/entry.cpp:8-12,
// include/hpc/rll/cuda/models/entry.h:11-21): three forward-only AlphaStar actor-critic inference helpers with the
// reference's `Fn(inputs, outputs)` convention and tensor order (src/models/actor_critic.cu:8-83).  Host-only C++.
#include "common.hpp"

namespace hpc_rll_ext {
namespace {

// inputs = [key_embeddings (B,E,D), sample_entity (B,) int64, entity_num (B,) int64];
// outputs = [autoregressive_embedding (B,D)], updated IN PLACE:
// ae[b] += key_embeddings[b, sample_entity[b]] unless sample_entity[b] == entity_num[b] (the "end" action).
void actor_critic_update_ae(const TensorList& in, const TensorList& out) {
    expect_len(in, 3, "actor_critic_update_ae inputs");
    expect_len(out, 1, "actor_critic_update_ae outputs");
    const Tensor& key = req(in[0], "key_embeddings");
    TORCH_CHECK(key.dim() == 3, "key_embeddings: expected (B,E,D), got ", key.sizes());
    const int64_t B = key.size(0), E = key.size(1), D = key.size(2);
    const at::Device dev = key.device();
    req(in[1], "sample_entity", dev, {B}, at::kLong);
    req(in[2], "entity_num", dev, {B}, at::kLong);
    req(out[0], "autoregressive_embedding", dev, {B, D});
    c10::DeviceGuard g(dev);
    check(hpc_rll_actor_critic_update_ae(fptr(key), iptr(in[1]), iptr(in[2]), fmut(out[0]), B, E, D, stream_of(dev)),
          "hpc_rll_actor_critic_update_ae");
}

// inputs = [lstm_ih (B,4H), lstm_hh (B,4H), bias (4H,)], outputs = [h (B,H) written, c (B,H) updated in place];
// gate order i,f,g,o (torch.nn.LSTM).
void actor_critic_lstm_activation(const TensorList& in, const TensorList& out) {
    expect_len(in, 3, "actor_critic_lstm_activation inputs");
    expect_len(out, 2, "actor_critic_lstm_activation outputs");
    const Tensor& ih = req(in[0], "lstm_ih");
    TORCH_CHECK(ih.dim() == 2 && ih.size(1) % 4 == 0, "lstm_ih: expected (B,4H), got ", ih.sizes());
    const int64_t B = ih.size(0), G = ih.size(1), H = G / 4;
    const at::Device dev = ih.device();
    req(in[1], "lstm_hh", dev, {B, G});
    req(in[2], "lstm_bias", dev);
    TORCH_CHECK(in[2].numel() == G, "lstm_bias: ", in[2].numel(), " elements, expected ", G);
    req(out[0], "lstm_hx", dev);
    req(out[1], "lstm_cx", dev);
    TORCH_CHECK(out[0].numel() == B * H && out[1].numel() == B * H, "lstm_hx / lstm_cx: expected B*H elements");
    c10::DeviceGuard g(dev);
    check(hpc_rll_actor_critic_lstm_activation(fptr(ih), fptr(in[1]), fptr(in[2]), fmut(out[0]), fmut(out[1]), B, H,
                                               stream_of(dev)),
          "hpc_rll_actor_critic_lstm_activation");
}

// inputs = [mat (B,E,H), vec (.., B, H) (any leading singleton dims), mask (B,E) bool]; outputs = [out (B,E)]:
// out = where(mask, (mat * vec[:,None,:]).sum(-1), -1e9) / 0.8.
void actor_critic_pre_sample(const TensorList& in, const TensorList& out) {
    expect_len(in, 3, "actor_critic_pre_sample inputs");
    expect_len(out, 1, "actor_critic_pre_sample outputs");
    const Tensor& mat = req(in[0], "mat");
    TORCH_CHECK(mat.dim() == 3, "mat: expected (B,E,H), got ", mat.sizes());
    const int64_t B = mat.size(0), E = mat.size(1), H = mat.size(2);
    const at::Device dev = mat.device();
    req(in[1], "vec", dev);
    TORCH_CHECK(in[1].numel() == B * H, "vec: ", in[1].sizes(), " does not hold B*H = ", B * H, " values");
    req(in[2], "mask", dev, {B, E}, at::kBool);
    req(out[0], "output", dev, {B, E});
    c10::DeviceGuard g(dev);
    check(hpc_rll_actor_critic_pre_sample(fptr(mat), fptr(in[1]), (const uint8_t*)in[2].const_data_ptr<bool>(),
                                          fmut(out[0]), B, E, H, -1e9f, 0.8f, stream_of(dev)),
          "hpc_rll_actor_critic_pre_sample");
}

}  // namespace
}  // namespace hpc_rll_ext

PYBIND11_MODULE(hpc_models, m) {
    using namespace hpc_rll_ext;
    m.doc() = "hpc_models: AlphaStar actor-critic inference helpers for MI355X (gfx950) -- compiled PyTorch-ROCm "
              "extension over the C ABI of libhpc_rll_hip.so (reference: src/models/entry.cpp:8-12)";
    bind_common(m);
    m.def("actor_critic_update_ae", &actor_critic_update_ae, "actor critic update autoregressive embedding (HIP)");
    m.def("actor_critic_lstm_activation", &actor_critic_lstm_activation, "actor critic lstm activation (HIP)");
    m.def("actor_critic_pre_sample", &actor_critic_pre_sample, "actor critic pre sample (HIP)");
}
