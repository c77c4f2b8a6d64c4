// wave.hpp -- wave64 device primitives for gfx950 (CDNA4).
//
// Re-design (not a translation) of the reference's include/hpc/rll/cuda/reduce.h:13-99 and
// basic_math.h:15-29: those assume 32-lane warps and a 32-slot static shared array that is reused
// without a barrier.  Here: 64-lane butterflies in registers, one LDS slot per
// wave, explicit barriers, and deterministic two-stage scalar reductions (no float atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hpc_rll {

constexpr int kWave = 64;

// ---- vector packs (V consecutive fp32 columns owned by one lane) ----------------------------
template <int V> struct Pack { float v[V]; };

typedef float vfloat2 __attribute__((ext_vector_type(2)));
typedef float vfloat4 __attribute__((ext_vector_type(4)));

// NT = nontemporal ("streaming") access: the line is not kept in L2/MALL.  Measured on MI355X: nontemporal
// STORES lift write-heavy streaming kernels from ~6.0 to ~6.9 TB/s (no write-allocate traffic competing with
// the reads); nontemporal LOADS help read-once streams when enough loads are in flight.
template <bool NT, class T> __device__ __forceinline__ T ld(const T* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <bool NT, class T> __device__ __forceinline__ void st(T* p, T x) {
    if (NT) __builtin_nontemporal_store(x, p);
    else *p = x;
}

template <int V, bool NT = false> struct PackIO;
template <bool NT> struct PackIO<1, NT> {
    static __device__ __forceinline__ Pack<1> load(const float* p) { return Pack<1>{{ld<NT>(p)}}; }
    static __device__ __forceinline__ void store(float* p, const Pack<1>& x) { st<NT>(p, x.v[0]); }
};
template <bool NT> struct PackIO<2, NT> {
    static __device__ __forceinline__ Pack<2> load(const float* p) {
        const vfloat2 t = ld<NT>(reinterpret_cast<const vfloat2*>(p));
        return Pack<2>{{t.x, t.y}};
    }
    static __device__ __forceinline__ void store(float* p, const Pack<2>& x) {
        vfloat2 t; t.x = x.v[0]; t.y = x.v[1];
        st<NT>(reinterpret_cast<vfloat2*>(p), t);
    }
};
template <bool NT> struct PackIO<4, NT> {
    static __device__ __forceinline__ Pack<4> load(const float* p) {
        const vfloat4 t = ld<NT>(reinterpret_cast<const vfloat4*>(p));
        return Pack<4>{{t.x, t.y, t.z, t.w}};
    }
    static __device__ __forceinline__ void store(float* p, const Pack<4>& x) {
        vfloat4 t; t.x = x.v[0]; t.y = x.v[1]; t.z = x.v[2]; t.w = x.v[3];
        st<NT>(reinterpret_cast<vfloat4*>(p), t);
    }
};
template <int V, bool NT = false> __device__ __forceinline__ Pack<V> load_pack(const float* p) {
    return PackIO<V, NT>::load(p);
}
template <int V, bool NT = false> __device__ __forceinline__ void store_pack(float* p, const Pack<V>& x) {
    PackIO<V, NT>::store(p, x);
}

// ---- wave-level reductions (all 64 lanes end with the result) --------------------------------
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x = fmaxf(x, __shfl_xor(x, m, 64));
    return x;
}
// ---- DPP-only sums (no LDS traffic).  After the steps for a group of GL lanes (8, 16, 32 or 64, aligned) the LAST
// lane of every group holds the group total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK,
                                                                     0xF, false));
}
template <int GL>
__device__ __forceinline__ float group_sum_last(float x) {
    x = dpp_add<0xB1, 0xF>(x);                  // quad_perm [1,0,3,2]
    x = dpp_add<0x4E, 0xF>(x);                  // quad_perm [2,3,0,1]
    x = dpp_add<0x141, 0xF>(x);                 // row_half_mirror: 8-lane sums in every lane
    if (GL >= 16) x = dpp_add<0x140, 0xF>(x);   // row_mirror: 16-lane row sums in every lane
    if (GL >= 32) x = dpp_add<0x142, 0xA>(x);   // row_bcast:15 -> rows 1 and 3 hold 32-lane sums
    if (GL >= 64) x = dpp_add<0x143, 0xC>(x);   // row_bcast:31 -> row 3 holds the wave total
    return x;
}
// Sum over the 32 lanes of a HALF wave (lanes sharing lane >> 5 -- the column index of a 32x32 matrix-core block);
// every lane of the half gets the total.  Four DPP steps (16-lane row sums in every lane) + one cross-row exchange.
__device__ __forceinline__ float half_sum_all(float x) {
    x = dpp_add<0xB1, 0xF>(x);    // quad_perm [1,0,3,2]
    x = dpp_add<0x4E, 0xF>(x);    // quad_perm [2,3,0,1]
    x = dpp_add<0x141, 0xF>(x);   // row_half_mirror
    x = dpp_add<0x140, 0xF>(x);   // row_mirror
    return x + __shfl_xor(x, 16, 64);
}
// reductions restricted to aligned groups of G lanes (G a power of two <= 64)
template <int G> __device__ __forceinline__ float group_sum(float x) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}
template <int G> __device__ __forceinline__ float group_max(float x) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) x = fmaxf(x, __shfl_xor(x, m, 64));
    return x;
}

// ---- workgroup reduction of K running sums -> K per-workgroup partials ------------------------
// Every thread passes its K partial sums; thread 0 returns with the K workgroup totals in out[].
// `lds` must hold K * (blockDim.x/64) floats.  Contains __syncthreads(): call from all threads.
template <int K>
__device__ __forceinline__ void block_sum(const float (&in)[K], float (&out)[K], float* lds) {
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float s = wave_sum(in[k]);
        if (lane == 0) lds[k * nw + w] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float s = 0.f;
            for (int i = 0; i < nw; ++i) s += lds[k * nw + i];  // fixed order: deterministic
            out[k] = s;
        }
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

}  // namespace hpc_rll
