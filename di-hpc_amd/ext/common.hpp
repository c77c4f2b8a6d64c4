// common.hpp -- shared host-side glue of the compiled PyTorch-ROCm extension modules
// (hpc_rl_utils, hpc_torch_utils_network, hpc_models).
//
// These modules are the "L2" of the reference (src/rl_utils/entry.cpp:8-39, src/torch_utils/network/entry.cpp:8-13,
// src/models/entry.cpp:8-12): pybind functions with the reference's names and `Fn(inputs, outputs, scalars)` calling
// convention.  Here they are HOST-ONLY C++ (g++, no device code, no hipify): they validate tensors, take outputs and
// scratch from torch's caching allocator, fetch torch's CURRENT HIP stream and call the torch-free C ABI of
// libhpc_rll_hip.so (include/hpc_rll_hip.h).  On top of the L2 functions each module exports the fused autograd ops
// (torch::autograd::Function) the hpc_rll.* Python modules call -- one pybind call per Module.forward, backward
// entirely inside the autograd engine.
//
// There is no CPU / eager fallback: a non-GPU tensor is a RuntimeError.
#pragma once

#include <torch/extension.h>

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include <algorithm>
#include <cstdint>
#include <initializer_list>
#include <mutex>
#include <optional>
#include <string>
#include <unordered_map>
#include <vector>

#include "hpc_rll_hip.h"

namespace hpc_rll_ext {

using at::Tensor;
using TensorList = std::vector<Tensor>;
using OptTensor = std::optional<Tensor>;
using OptList = std::vector<OptTensor>;   // python lists that may hold None
namespace ag = torch::autograd;

constexpr int kAbiVersion = 6;   // 3: + kernel timing, packed / grouped pad, last GAE configuration (round 3); 4: + lstm_*_y (round 4); 5: tune table, retired keys (round 5); 6: + iqn_*_bnt (round 6)

// torch's current stream of `dev` as the void* the C ABI takes.  (Tensors of a ROCm build carry DeviceType::CUDA
// while c10::hip::HIPStream carries DeviceType::HIP; the underlying thread-local current stream is the same.)
inline void* stream_of(const at::Device& dev) {
    return (void*)c10::hip::getCurrentHIPStream(dev.index()).stream();
}

// C-ABI status -> the reference's error behaviour (RuntimeError, cf. include/hpc/rll/cuda/status.h:19-28).
inline void check(int status, const char* what) {
    TORCH_CHECK(status == 0, what, ": ", hpc_rll_status_string(status), " (status ", status, ")");
}

// Validation the reference omits (status.h:15-17 defines CHECK_* but never uses them).
inline const Tensor& req(const Tensor& t, const char* name, at::ScalarType dtype = at::kFloat) {
    TORCH_CHECK(t.defined(), name, ": expected a tensor, got None");
    TORCH_CHECK(t.is_cuda(), name, ": must live on a GPU (hpc_rll has no CPU path)");
    TORCH_CHECK(t.scalar_type() == dtype, name, ": dtype ", t.scalar_type(), ", expected ", dtype);
    TORCH_CHECK(t.is_contiguous(), name, ": must be contiguous");
    return t;
}
inline const Tensor& req(const Tensor& t, const char* name, const at::Device& dev, at::ScalarType dtype = at::kFloat) {
    req(t, name, dtype);
    TORCH_CHECK(t.device() == dev, name, ": on ", t.device(), ", expected ", dev);
    return t;
}
inline const Tensor& req(const Tensor& t, const char* name, const at::Device& dev, std::initializer_list<int64_t> shape,
                         at::ScalarType dtype = at::kFloat) {
    req(t, name, dev, dtype);
    TORCH_CHECK(t.sizes() == at::IntArrayRef(shape.begin(), shape.size()), name, ": shape ", t.sizes(), ", expected ",
                at::IntArrayRef(shape.begin(), shape.size()));
    return t;
}
inline void req_opt(const OptTensor& t, const char* name, const at::Device& dev, std::initializer_list<int64_t> shape,
                    at::ScalarType dtype = at::kFloat) {
    if (t.has_value() && t->defined()) req(*t, name, dev, shape, dtype);
}
inline bool has(const OptTensor& t) { return t.has_value() && t->defined(); }

inline const float* fptr(const Tensor& t) { return t.defined() ? t.const_data_ptr<float>() : nullptr; }
inline float* fmut(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
inline const float* fptr(const OptTensor& t) { return has(t) ? t->const_data_ptr<float>() : nullptr; }
inline float* fmut(const OptTensor& t) { return has(t) ? t->data_ptr<float>() : nullptr; }
inline const int64_t* iptr(const Tensor& t) { return t.const_data_ptr<int64_t>(); }

inline Tensor new_f32(at::IntArrayRef shape, const at::Device& dev) {
    return at::empty(shape, at::TensorOptions().dtype(at::kFloat).device(dev));
}
inline Tensor undef() { return Tensor(); }

inline int to_int(int64_t v, const char* name) {
    TORCH_CHECK(v >= 0 && v <= INT32_MAX, name, ": ", v, " does not fit the kernels' 32-bit size");
    return (int)v;
}

// Scalar gradient arriving from autograd -> a contiguous (1,) fp32 device tensor the kernels read g[0] from.
inline Tensor grad1(const Tensor& g, const at::Device& dev, const char* name) {
    Tensor r = g.reshape({1});
    if (!r.is_contiguous()) r = r.contiguous();
    return req(r, name, dev);
}

// `n` expected tensors in a python list.
template <class L> inline void expect_len(const L& v, size_t n, const char* what) {
    TORCH_CHECK(v.size() == n, what, ": expected ", n, " tensors, got ", v.size());
}

// 1/(local count) unless the data-parallel caller passed 1/(GLOBAL count) (hpc_rll.dist.loss_scale).
inline float loss_scale(std::optional<double> scale, int64_t local_count) {
    if (scale.has_value() && *scale > 0.0) return (float)*scale;
    return (float)(1.0 / (double)(local_count > 0 ? local_count : 1));
}

// State a reference-convention L2 forward must hand to its backward.  The reference's L1 modules pass module-owned
// scratch buffers positionally (e.g. hpc_rll/rl_utils/vtrace.py:17-27) and their backward lists do not carry what
// this library's recompute-in-backward kernels need (the logits, ONE workspace).  The forward therefore parks the
// tensors under the address of one of the reference's own scratch buffers that also appears in the backward list,
// and the backward picks them up.  An entry lives until the next forward with the same buffer replaces it
// (backward may run more than once: retain_graph); entries whose module buffer has been freed are purged at every put()
// (modules created and dropped in a loop pin nothing), and the table keeps at most the kCap most recently written ones.
class SavedByBuffer {
  public:
    static constexpr size_t kCap = 4096;   // LIVE module buffers with a pending backward (entries of freed buffers are purged in put())
    struct Entry {
        std::vector<Tensor> tensors;
        double scalar = 0.0;
        uint64_t seed = 0;
        uint64_t stamp = 0;
        // identity of the key buffer and the state of the parked tensors at forward time: an address that
        // was freed and handed to another tensor must not resolve to the old module's state, and an input modified in
        // place between forward and backward is an error here exactly as on the native (save_for_backward) path
        c10::weak_intrusive_ptr<c10::StorageImpl> key_storage{c10::intrusive_ptr<c10::StorageImpl>()};
        std::vector<uint32_t> versions;
    };
    void put(const Tensor& key, Entry e) {
        std::lock_guard<std::mutex> lk(mu_);
        e.stamp = ++clock_;
        e.key_storage = c10::weak_intrusive_ptr<c10::StorageImpl>(key.storage().getWeakStorageImpl());
        e.versions.clear();
        for (const Tensor& t : e.tensors) e.versions.push_back(t.defined() ? t._version() : 0);
        // entries whose module buffer is gone can never be asked for again: release what they pin
        for (auto it = map_.begin(); it != map_.end();) {
            if (it->second.key_storage.expired()) it = map_.erase(it); else ++it;
        }
        map_[key.data_ptr()] = std::move(e);
        if (map_.size() > kCap) {   // drop the oldest half (rare: amortised O(1))
            std::vector<std::pair<uint64_t, void*>> order;
            order.reserve(map_.size());
            for (auto& kv : map_) order.emplace_back(kv.second.stamp, kv.first);
            std::sort(order.begin(), order.end());
            for (size_t i = 0; i < order.size() / 2; ++i) map_.erase(order[i].second);
        }
    }
    Entry get(const Tensor& key, const char* what) {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = map_.find(key.data_ptr());
        TORCH_CHECK(it != map_.end(), what, ": no forward state for this scratch buffer (call the Forward with the "
                    "same module buffers first)");
        const auto live = it->second.key_storage.lock();
        TORCH_CHECK(live && live.get() == key.storage().unsafeGetStorageImpl(), what, ": the scratch buffer at this address is "
                    "not the one the Forward was called with (freed and re-allocated?)");
        for (size_t i = 0; i < it->second.tensors.size(); ++i) {
            const Tensor& t = it->second.tensors[i];
            TORCH_CHECK(!t.defined() || t._version() == it->second.versions[i], what, ": a tensor the backward pass re-reads "
                        "was modified in place after the Forward");
        }
        return it->second;
    }
  private:
    std::mutex mu_;
    uint64_t clock_ = 0;
    std::unordered_map<void*, Entry> map_;
};

inline std::string lib_info() {
    return std::string("libhpc_rll_hip.so ABI ") + std::to_string(hpc_rll_abi_version());
}

inline void check_abi() {
    TORCH_CHECK(hpc_rll_abi_version() == kAbiVersion, "libhpc_rll_hip.so ABI ", hpc_rll_abi_version(), " != expected ",
                kAbiVersion, "; rebuild with `python di-hpc_amd/build.py`");
}

// Common bindings every module carries (tests and tuning scripts use them).
// `n` consecutive floats of `buf` starting at `offset` as an INDEPENDENT tensor over the same storage (not a view in
// autograd's sense): several outputs of one autograd node may alias one buffer, and the caller may still modify each of
// them in place (`loss += ...`) -- narrow() views would raise "Output 0 of ...Backward is a view and is being modified
// inplace ... returns multiple views" (the reference returns separate module buffers).
inline at::Tensor alias_of(const at::Tensor& buf, int64_t offset, int64_t n) {
    at::Tensor t = at::empty({0}, buf.options());
    t.set_(buf.storage(), buf.storage_offset() + offset, {n}, {1});
    return t;
}

inline void bind_common(pybind11::module_& m) {
    check_abi();
    m.def("abi_version", []() { return hpc_rll_abi_version(); });
    m.def("tune_set", [](int key, int value) { check(hpc_rll_tune_set(key, value), "tune_set"); },
          "hpc_rll_tune_set(key, value): experiment knobs documented in include/hpc_rll_hip.h");
    m.def("status_string", [](int s) { return std::string(hpc_rll_status_string(s)); });
}

}  // namespace hpc_rll_ext
