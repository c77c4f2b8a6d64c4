// lstm_mid.hpp -- persistent MID-BATCH kernels of the LayerNorm-LSTM: forward, and a backward (below) (included by lstm.hip only).
//
// Regime: 5 <= B <= 256, 64 <= H <= 1024, H % 16 == 0 -- the batches RL actors and small learners run (the reference's own
// test is B = 3: tests/test_lstm.py:10-16; its per-step cost model is src/torch_utils/network/lstm.cu:145-161: one SGEMM,
// one LayerNorm and one activation launch per step).  Measured before this kernel (profiles/r04_lstm_mid_table.json): the
// two-launch step (split-K product + cell kernel) costs 13.5-53 us per step where the matrix pipe needs 0.1-13.6 us -- the
// step is bound by launch boundaries and by streaming Wh (16 MB at H = 1024) from L2 / HBM every step.
//
// Here ONE kernel per layer walks all S steps (the idea of lstm_persist.hpp, with the product on the matrix cores):
//   * workgroup w (1024 threads = 16 waves, one per CU) owns hidden units 4w .. 4w+3, i.e. 16 gate columns of Wh, which stay
//     in LDS for the whole sequence (H x 16 floats = 64 KB at H = 1024) in the order the matrix core's B operand wants
//     them: Wl[k / 4][column][k % 4], one conflict-free ds_read_b128 per four k;
//   * per step the workgroup computes its (B, 16) slice of h_{s-1} @ Wh with v_mfma_f32_16x16x4_f32: wave (mb, kq) takes the
//     16-row block mb and the k slice kq (16 waves = mbp row blocks x ks k slices), reads its A operand straight from
//     h_{s-1} in global memory (L2: every workgroup reads all of it) and leaves a 16 x 16 partial tile in LDS;
//   * thread (row, unit) adds the k slices of its four gates (a fixed order: deterministic), keeps c in a register;
//   * two exchanges per step between ALL workgroups: (1) LayerNorm partials -- every workgroup sends (mean, M2) of its 16
//     columns per row as {value, tag} words to ONE combiner workgroup per row, which sends (mean, rstd) back the same way
//     -- and (2) h_s itself: a slot per step, write-through stores, a flag word per workgroup carrying the step number (the
//     waiting wave reads all flags in one poll: no atomics, no counter), ordinary loads by the readers.  No cache-wide
//     fences anywhere (see "Exchange protocol" below).
// Co-residency, bounded waits and the timeout protocol are lstm_persist.hpp's (persist_runtime_ready, persist_resident_t,
// persist_poll_failed).  The saved-for-backward tensors (hw, gates, c, hseq, stats) are written exactly as the step kernels
// write them, so either backward (the step kernels, or lstm_mid_bwd_kernel below) can follow.
// Measured history of the protocol, phase times and what bounds the step: DESIGN.md 4.6, profiles/r04_lstm_mid_phases.txt,
// profiles/r04_bcast_micro.txt, profiles/r04_lstm_mid_bwd.txt.
#pragma once

namespace hpc_rll {
int g_lstm_mid = 2;       // hpc_rll_tune_set key 29: 0 off, 1 one stream, 2 two streams where they fit
constexpr int g_lstm_mid_rep = 8;   // replicas of the words every workgroup polls (flags, final row statistics): 256 pollers on one line cost 4.4 us per exchange
int g_lstm_mid_bwd = 1;   // hpc_rll_tune_set key 33: the persistent mid-batch BACKWARD: 0 off, 1 where it pays (B <= 32), 2 every mid-batch shape
namespace {

constexpr int kMidMaxB = 256;
constexpr int kMidMaxRep = 32;
#ifndef HPC_RLL_MID_CHUNK
#define HPC_RLL_MID_CHUNK 16
#endif
constexpr int kMidChunk = HPC_RLL_MID_CHUNK;

struct MidFwd {
    const float *xw, *wh, *bias, *gamma, *beta, *h0, *c0;
    float *hw, *gates, *c, *hseq, *stats;
    float* hx;                     // [S][nwg][rows][4]: h_s as the workgroups exchange it (unit quads; see the kernel)
    u64* part_t;                   // [rows][nwg][2] {value, tag}: (mean, M2) of a workgroup's 16 columns, tag = step + 1
    u64* fin_t;                    // [nrep][256][2] {value, tag}: (mean, rstd) of the whole row, from the row's combiner
    unsigned* flag_h;              // [nrep][256]: steps whose h the workgroup has published
    // (words that EVERY workgroup polls are replicated: 256 pollers on one line cost 4.4 us per exchange where a single
    // reader sees a store after 0.6 us -- tests/tools/micro/bcast.hip, pingpong.hip; workgroup w reads replica w % nrep)
    int S, Btot, Bs /* rows per stream */, H, nwg, mbp /* 16-row blocks (a power of two) */, ks /* k slices; mbp * ks <= waves */, nrep;
    u64* prof;
};

// Exchange protocol (no cache-wide fences -- the first version of this kernel used agent-scope release / acquire fences
// like lstm_block.hpp and spent 30 of its 40 us per step in them: a release writes back EVERY dirty line of the XCD's L2, an
// acquire invalidates the whole L2, so 32 workgroups per XCD kept flushing and refetching each other's data):
//   * exchanged data goes to addresses that are written ONCE per launch (a slot per step) with write-through stores
//     (agent scope: sc1), whole 128-byte lines per writer; the writer's waves wait for their own stores (s_waitcnt
//     vmcnt(0)) before the workgroup barrier, then one thread stores the workgroup's flag = step number;
//   * the first wave of a waiting workgroup polls all flags with agent-scope loads (one round of loads per poll);
//   * readers then use ORDINARY loads: the lines were never read before in this launch (nothing stale in L1 / L2; a launch
//     starts with clean caches), so the 32 workgroups of an XCD share ONE fetch of every line through their L2.
__device__ __forceinline__ void mid_store1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mid_store2(float* p, float x, float y) {
    __hip_atomic_store(reinterpret_cast<u64*>(p), ((u64)__float_as_uint(y) << 32) | (u64)__float_as_uint(x),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16 bytes, agent scope (sc1): the buffer form of the store carries the cache-policy bits (0x10 = sc1 on gfx940+)
// (`base` must be wave-uniform: a per-lane descriptor makes the compiler loop over the lanes)
__device__ __forceinline__ void mid_store4(float* base, unsigned byte_off, vfloat4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(blk_u32x4, v), blk_rsrc(base), (int)byte_off, 0, 16);
}
// One failed poll of the mid-batch kernels: lstm_persist.hpp's bounded wait (abort word, limit, host status) with a shorter nap --
// there the co-resident waves need the memory queue, here the rest of the workgroup sits at a barrier and a poll round trip is
// the step's critical path: 64 cycles instead of 512.
__device__ __forceinline__ void mid_poll_failed(long& spins) {
    if (__builtin_expect(((++spins) & 1023) == 0, 0)) {
        if (__hip_atomic_load(&g_persist_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) __builtin_amdgcn_endpgm();
        if (spins >= g_persist_spin_limit) {
            __hip_atomic_store(&g_persist_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned* hs = g_persist_host_status;
            if (hs) __hip_atomic_store(hs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __builtin_amdgcn_endpgm();
        }
    }
    __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void mid_publish(unsigned* flags, int nrep, unsigned value) {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's write-through stores have been acknowledged
    __syncthreads();
    if ((int)threadIdx.x < nrep)
        __hip_atomic_store(flags + threadIdx.x * 256 + blockIdx.x, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every workgroup's flag >= target (the first wave polls: all flags in one round of loads)
__device__ __forceinline__ void mid_wait(const unsigned* flags, int nwg, unsigned target) {
    if (threadIdx.x < 64) {
        const int l = threadIdx.x;
        long spins = 0;
        while (true) {
            unsigned v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                v[i] = __hip_atomic_load(flags + (l + 64 * i < nwg ? l + 64 * i : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
#pragma unroll
            for (int i = 0; i < 4; ++i) ok = ok && (l + 64 * i >= nwg || v[i] >= target);
            if (__all(ok)) break;
            mid_poll_failed(spins);
        }
    }
    __syncthreads();
}
__device__ __forceinline__ float quad_sum(float x) {   // over the four lanes of a quad, every lane gets the total
    x = dpp_add<0xB1, 0xF>(x);
    return dpp_add<0x4E, 0xF>(x);
}
// 4 x 4 transpose inside a quad: lane L, element i  <->  lane i, element L (two DPP exchanges per element pair, no LDS).
// The saved tensors are (row, gate, unit)-ordered with FOUR units per workgroup: thread (row, jj) owns unit jj of every gate,
// but 16 contiguous bytes in memory are the four units of ONE gate.  With this, lane jj moves gate jj's 16 bytes in one
// access and hands the units to their owners -- a quarter of the vector-memory instructions (issuing them 4 bytes at a
// time, 16 distant rows per instruction, cost the backward 7.5 us per step at B = 64).
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ void quad_transpose(float (&a)[4], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
#pragma unroll
    for (int p = 0; p < 4; p += 2) {
        const float r = dpp_mov<0xB1>(b0 ? a[p] : a[p + 1]);
        if (b0) a[p] = r; else a[p + 1] = r;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float r = dpp_mov<0x4E>(b1 ? a[p] : a[p + 2]);
        if (b1) a[p] = r; else a[p + 2] = r;
    }
}
__device__ __forceinline__ void quad_load4(const float* p, float (&a)[4], int lane) {   // p: this lane's GATE row piece (16 bytes)
    const vfloat4 v = *reinterpret_cast<const vfloat4*>(p);
    a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
    quad_transpose(a, lane);
}
__device__ __forceinline__ void quad_store4(float* p, const float (&a)[4], int lane) {
    float t[4] = {a[0], a[1], a[2], a[3]};
    quad_transpose(t, lane);
    *reinterpret_cast<vfloat4*>(p) = vfloat4{t[0], t[1], t[2], t[3]};
}

// STREAMS.  Batch rows are independent sequences, so the batch can be cut into `gridDim.y` streams that run the recurrence
// independently (own flags, tags and slots): with two streams of NW = 8 waves a CU holds TWO workgroups (one per stream,
// each with its own copy of a Wh slice: 2 x 78 KB of LDS at H = 1024), and while one stream sits in an exchange the other
// stream's product has the matrix pipe -- the exchanges (3-4 memory trips per step) are what bounds a single stream.
template <int NW>
__global__ __launch_bounds__(64 * NW, 4) void lstm_mid_fwd_kernel(MidFwd a) {
    extern __shared__ float smem[];
    constexpr int NT = 64 * NW;
    const int H = a.H, G = 4 * H, nwg = a.nwg, mbp = a.mbp, ks = a.ks, Btot = a.Btot;
    const int Bp = 16 * mbp;
    const int strm = blockIdx.y, row0 = strm * a.Bs;
    const int B = Btot - row0 < a.Bs ? Btot - row0 : a.Bs;   // rows of this stream
    a.flag_h += (size_t)strm * kMidMaxRep * 256;
    a.fin_t += (size_t)strm * kMidMaxRep * 512;
    a.part_t += (size_t)strm * Bp * nwg * 2;
    a.hx += (size_t)strm * a.S * nwg * Bp * 4;
    if (strm != 0) a.prof = nullptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* Wl = smem;                          // [H / 4][16][4]: column n = gate * 4 + unit
    float* pre = Wl + (size_t)H * 16;          // [NW waves][16 rows][16 columns]
    float* lnst = pre + NW * 256;              // [rows][2]: mean, rstd of the h-branch rows
    const int wg = (int)blockIdx.x, j0 = wg * 4;
    for (int e = tid; e < H * 16; e += NT) {
        const int n = e & 15, k = e >> 4;
        Wl[(((k >> 2) * 16 + n) << 2) + (k & 3)] = a.wh[(size_t)k * G + (n >> 2) * H + j0 + (n & 3)];
    }
    // cell ownership: thread (row, jj) keeps c of unit j0 + jj of batch row `row` in a register
    const int row = tid >> 2, jj = tid & 3, cj = j0 + jj;
    const bool cell = row < B;
    float creg = cell ? a.c0[(size_t)(row0 + row) * H + cj] : 0.f;
    // product role of this wave: 16-row block mb, k slice kq.  A operand of lane (r, j), iteration i: the four k
    // kq * Kw + 16 i + 4 j ... + 3 of row 16 mb + r = ONE unit quad: 16 bytes of h0 (row major) or of the exchange slot of
    // the workgroup that owns the quad
    const int mb = wave & (mbp - 1), kq = wave / mbp;
    const bool active = kq < ks;
    const int Kw = H / ks, n16 = Kw / 16;
    int ar = 16 * mb + (lane & 15);
    if (ar >= B) ar = B - 1;                   // rows past the batch repeat the last one (results unused)
    const int aj = lane >> 4;
    const int kq_a = active ? kq : 0;
    const size_t a0_off = (size_t)(row0 + ar) * H + (size_t)kq_a * Kw + 4 * aj;                  // h0: + 16 floats per iteration
    const size_t ax_off = ((size_t)(kq_a * (Kw / 4) + aj) * Bp + ar) * 4;                // hx: + 4 workgroups per iteration
    const size_t ax_step = (size_t)16 * Bp;
    const float* bp = Wl + ((size_t)(kq_a * (Kw / 4) + aj) * 16 + (lane & 15)) * 4;
    const float inv_g = 1.f / (float)G, inv_n = 1.f / (float)nwg;
    const size_t slot_h = (size_t)nwg * Bp * 4;
    float* cst = lnst + 2 * kMidMaxB;          // [3][16]: gamma_x, gamma_h, beta_x + beta_h + bias of the 16 owned columns
    float* cmb = cst + 48;                     // [2 * nwg]: a row's partials while its combiner reduces them
    if (tid < 16) {
        const int col = (tid >> 2) * H + j0 + (tid & 3);
        cst[tid] = a.gamma[col];
        cst[16 + tid] = a.gamma[G + col];
        cst[32 + tid] = (a.beta[col] + a.beta[G + col]) + a.bias[col];
    }
    __syncthreads();

    u64 tprev_ = a.prof ? wall_clock64() : 0;
    for (int s = 0; s < a.S; ++s) {
        const uint32_t tag = (uint32_t)s + 1u;
        if (s > 0) mid_wait(a.flag_h + (wg % a.nrep) * 256, nwg, (unsigned)s);   // h_{s-1} complete
        HPC_RLL_TICK(0)
        // ---- this workgroup's 16 columns of h_{s-1} @ Wh
        if (active) {
            vfloat4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float* ap = s == 0 ? a.h0 + a0_off : a.hx + (size_t)(s - 1) * slot_h + ax_off;
            const size_t astep = s == 0 ? (size_t)16 : ax_step;
            for (int i0 = 0; i0 < n16; i0 += kMidChunk) {   // 16-byte loads in flight per lane: one memory round trip per chunk
                vfloat4 av[kMidChunk];
#pragma unroll
                for (int u = 0; u < kMidChunk; ++u) av[u] = *reinterpret_cast<const vfloat4*>(ap + astep * (i0 + u < n16 ? i0 + u : i0));
                if (a.prof) {   // profiling runs only: the time until the operand has arrived, apart from the matrix instructions
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                    HPC_RLL_TICK(6)
                }
#pragma unroll
                for (int u = 0; u < kMidChunk; ++u)
                    if (i0 + u < n16) {
                        const vfloat4 bv = *reinterpret_cast<const vfloat4*>(bp + (size_t)(i0 + u) * 256);
                        // two accumulators: the instruction's dependent latency (40 cycles) is longer than its issue (32)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bv.x, acc, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bv.y, acc1, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bv.z, acc, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bv.w, acc1, 0, 0, 0);
                    }
            }
            acc += acc1;
            float* pw = pre + (size_t)wave * 256 + (4 * aj) * 16 + (lane & 15);   // D[row 4 * (lane >> 4) + r][column lane & 15]
#pragma unroll
            for (int r = 0; r < 4; ++r) pw[r * 16] = acc[r];
        }
        float xv[4] = {0.f, 0.f, 0.f, 0.f}, mx = 0.f, rx = 0.f;
        if (cell) {   // x branch of this step (in flight during the statistics exchange)
            quad_load4(a.xw + ((size_t)s * Btot + row0 + row) * G + jj * H + j0, xv, lane);   // (lane jj fetches gate jj)
            const vfloat2 st = *reinterpret_cast<const vfloat2*>(a.stats + ((size_t)s * Btot + row0 + row) * 4);
            mx = st.x;
            rx = st.y;
        }
        __syncthreads();
        HPC_RLL_TICK(1)
        // ---- k slices summed in slice order; LayerNorm partials of the 16 columns -> the row's combiner
        float p[4] = {0.f, 0.f, 0.f, 0.f};
        if (tid < 4 * Bp) {
            const float* pr = pre + (size_t)((row >> 4) * 16 + (row & 15)) * 16 + jj;
            for (int q = 0; q < ks; ++q) {
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) p[gg] += pr[(size_t)q * mbp * 256 + gg * 4];
            }
        }
        {
            const float s1 = quad_sum((p[0] + p[1]) + (p[2] + p[3]));
            const float ml = s1 * (1.f / 16.f);
            float d2 = 0.f;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) d2 += (p[gg] - ml) * (p[gg] - ml);
            d2 = quad_sum(d2);
            if (cell && jj == 0) {
                u64* pt = a.part_t + ((size_t)row * nwg + wg) * 2;
                xchg_put(pt, ml, tag);
                xchg_put(pt + 1, d2, tag);
            }
        }
        HPC_RLL_TICK(2)
        // ---- rows wg, wg + nwg, ...: this workgroup combines them.  The partials are {value, tag} words: the data word is its
        // own ready flag (one store -> load trip, no barrier); every partial covers 16 columns, so
        //   mean = sum(ml) / n,   M2 = sum(d2) + 16 sum((ml - mean)^2)
        for (int r = wg; r < B; r += nwg) {
            if (tid < 2 * nwg) {
                const u64* src = a.part_t + (size_t)r * nwg * 2 + tid;
                long spins = 0;
                u64 w;
                while ((uint32_t)((w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != tag)
                    mid_poll_failed(spins);
                cmb[tid] = __uint_as_float((uint32_t)w);
            }
            __syncthreads();
            if (tid < 64) {
                float t1 = 0.f, td = 0.f;
                for (int w = tid; w < nwg; w += 64) { t1 += cmb[2 * w]; td += cmb[2 * w + 1]; }
                const float mean = wave_sum(t1) * inv_n;
                float t2 = 0.f;
                for (int w = tid; w < nwg; w += 64) t2 = fmaf(cmb[2 * w] - mean, cmb[2 * w] - mean, t2);
                const float m2 = wave_sum(td) + 16.f * wave_sum(t2);
                const float rstd = rsqrtf(m2 * inv_g + kLnEps);
                if (tid < a.nrep) {
                    xchg_put(a.fin_t + (size_t)tid * 512 + 2 * r, mean, tag);
                    xchg_put(a.fin_t + (size_t)tid * 512 + 2 * r + 1, rstd, tag);
                }
                if (tid == 0) *reinterpret_cast<vfloat2*>(a.stats + ((size_t)s * Btot + row0 + r) * 4 + 2) = vfloat2{mean, rstd};
            }
            __syncthreads();
        }
        HPC_RLL_TICK(3)
        // ---- every workgroup: (mean, rstd) of all rows
        if (tid < 2 * B) {
            long spins = 0;
            u64 w;
            const u64* src = a.fin_t + (size_t)(wg % a.nrep) * 512 + tid;
            while ((uint32_t)((w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != tag)
                mid_poll_failed(spins);
            lnst[tid] = __uint_as_float((uint32_t)w);
        }
        __syncthreads();
        HPC_RLL_TICK(4)
        // ---- cell
        float ig = 0.f, fg = 0.f, og = 0.f, ug = 0.f, hval = 0.f;
        if (cell) {
            const vfloat2 ln = *reinterpret_cast<const vfloat2*>(lnst + 2 * row);
            float av[4];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
                av[gg] = ((xv[gg] - mx) * rx * cst[gg * 4 + jj] + (p[gg] - ln.x) * ln.y * cst[16 + gg * 4 + jj]) + cst[32 + gg * 4 + jj];
            ig = blk_sigmoid<true>(av[0]);
            fg = blk_sigmoid<true>(av[1]);
            og = blk_sigmoid<true>(av[2]);
            ug = blk_tanh<true>(av[3]);
            creg = fg * creg + ig * ug;
            hval = og * blk_tanh<true>(creg);
            if (s + 1 < a.S) mid_store1(a.hx + (size_t)s * slot_h + ((size_t)wg * Bp + row) * 4 + jj, hval);
        }
        if (s + 1 < a.S) mid_publish(a.flag_h, a.nrep, (unsigned)(s + 1));
        if (cell) {   // saved for the backward / the caller: nobody in this launch waits for these
            const size_t r = (size_t)s * Btot + row0 + row;
            a.hseq[r * H + cj] = hval;
            const float gv[4] = {ig, fg, og, ug};
            quad_store4(a.gates + r * G + jj * H + j0, gv, lane);
            quad_store4(a.hw + r * G + jj * H + j0, p, lane);
            a.c[r * H + cj] = creg;
        }
        HPC_RLL_TICK(5)
    }
}

struct MidCfg { int ns /* streams */, nw /* waves per workgroup */, bs /* rows per stream */, mbp, ks, nwg; size_t lds; };
// Shape-only (the workspace layout depends on it)
inline bool lstm_mid_shape(int B, int H) { return B >= 5 && B <= kMidMaxB && H >= 64 && H <= 1024 && H % 16 == 0; }
inline size_t mid_lds(int H, int nw) { return ((size_t)H * 16 + (size_t)nw * 256 + 2 * kMidMaxB + 48 + 512) * sizeof(float); }
inline MidCfg mid_cfg(int B, int H, int ns) {
    MidCfg c;
    c.ns = ns;
    c.nw = ns == 2 ? 8 : 16;
    c.bs = (B + ns - 1) / ns;
    const int mb = (c.bs + 15) / 16;
    c.mbp = 1;
    while (c.mbp < mb) c.mbp *= 2;
    c.ks = c.nw / c.mbp;
    while (c.ks > 1 && (H / 16) % c.ks) c.ks /= 2;    // every k slice a multiple of 16 (four lanes x one 16-byte load)
    c.nwg = H / 4;
    c.lds = mid_lds(H, c.nw);
    return c;
}
// streams for this shape under the current knobs (key 29: 1 = one stream, 2 = two when two workgroups fit a CU's LDS)
// Measured (tests/tools/r04_lstm_mid_ab.py, profiles/r04_lstm_mid_ab.json): two streams win where the product is a large
// part of the step (H <= 512: 27.7 -> 20.7 us at B = 256, 11.4 -> 9.6 at B = 64) and tie or lose 0.3-0.7 us at H = 1024,
// where a stream's step is its chain of exchanges either way.
inline int mid_streams(int B, int H) { return g_lstm_mid >= 2 && H <= 512 && 2 * mid_lds(H, 8) <= 160 * 1024 ? 2 : 1; }
// Shapes the kernel is faster at than the two-launch step (same measurement): everything but B * H > 128 Ki with H > 512
// (B = 256, H = 1024: 41 against 40 us -- every workgroup reads the whole 1 MB h_{s-1}, 14 us, before a 14 us product).
inline bool mid_pays(int B, int H) { return (long)B * H <= 131072 || H <= 512; }
// workspace per stream: [flags: 32 x 256 words][fin_t: 32 x 256 x 2 u64][part_t: rows * nwg * 2 u64] ... then, line-aligned,
// [hx: a slot per step].  Sized for the larger of the one- and the two-stream layout (shape-only).
constexpr size_t kMidFlagFloats = (size_t)kMidMaxRep * 256, kMidFinFloats = (size_t)kMidMaxRep * 256 * 4;
inline size_t mid_ws_floats_ns(int S, int B, int H, int ns) {
    const MidCfg c = mid_cfg(B, H, ns);
    const size_t rows = (size_t)16 * c.mbp;
    return ns * (kMidFlagFloats + kMidFinFloats + rows * c.nwg * 4) + 32 + (size_t)ns * S * c.nwg * rows * 4;
}
inline size_t mid_ws_floats(int S, int B, int H) {
    if (!lstm_mid_shape(B, H) || S <= 0) return 0;
    const size_t a = mid_ws_floats_ns(S, B, H, 1), b = mid_ws_floats_ns(S, B, H, 2);
    return a > b ? a : b;
}
inline bool mid_fwd_ok(int B, int H, hipStream_t st) {
    if (!g_lstm_mid || !g_lstm_persist || !lstm_mid_shape(B, H) || !mid_pays(B, H) || !persist_runtime_ready(st)) return false;
    const MidCfg c = mid_cfg(B, H, mid_streams(B, H));
    return c.ns == 2 ? persist_resident_t(lstm_mid_fwd_kernel<8>, 512, 2 * c.nwg, c.lds)
                     : persist_resident_t(lstm_mid_fwd_kernel<16>, 1024, c.nwg, c.lds);
}
template <int NW>
inline int launch_mid_fwd_t(const MidCfg& c, const MidFwd& a, hipStream_t st) {
    if (c.lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void*)lstm_mid_fwd_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds);
        if (e != hipSuccess) return (int)e;
    }
    persist_chain_before(st);
    hipLaunchKernelGGL(lstm_mid_fwd_kernel<NW>, dim3(c.nwg, c.ns), dim3(64 * NW), c.lds, st, a);
    persist_chain_after(st);
    return 0;
}
inline int launch_mid_fwd(MidFwd a, float* ws_mid, int layer, hipStream_t st) {
    const MidCfg c = mid_cfg(a.Btot, a.H, mid_streams(a.Btot, a.H));
    a.Bs = c.bs;
    a.nwg = c.nwg;
    a.mbp = c.mbp;
    a.ks = c.ks;
    const size_t rows = (size_t)16 * c.mbp;
    const size_t polled = c.ns * (kMidFlagFloats + kMidFinFloats + rows * c.nwg * 4);   // floats: flags, fin_t, part_t of all streams
    a.nrep = g_lstm_mid_rep < 1 ? 1 : g_lstm_mid_rep > kMidMaxRep ? kMidMaxRep : g_lstm_mid_rep;
    a.flag_h = reinterpret_cast<unsigned*>(ws_mid);
    a.fin_t = reinterpret_cast<u64*>(ws_mid + c.ns * kMidFlagFloats);
    a.part_t = a.fin_t + c.ns * kMidFinFloats / 2;
    a.hx = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws_mid + polled) + 127) & ~(uintptr_t)127);   // a writer's block = whole lines
    if (hipMemsetAsync(ws_mid, 0, polled * sizeof(float), st) != hipSuccess) return last_error();   // flags and tags
    a.prof = persist_prof();
    const int rc = c.ns == 2 ? launch_mid_fwd_t<8>(c, a, st) : launch_mid_fwd_t<16>(c, a, st);
    if (rc) return rc;
    persist_prof_report("mid-batch fwd: wait_h product(after operand arrival) partials combine(own rows) wait_stats cell+publish operand_arrival", layer, a.S, st);
    return last_error();
}


// ================================================================================================== backward
// The backward recurrence of a layer in ONE launch for the same shapes (tune key 33): the structure of lstm_persist.hpp's
// backward kernel (thread (row, unit) keeps dh and dc of its unit in registers; per step the four LayerNorm-adjoint row sums
// and dHW_s are exchanged between all workgroups) on the forward kernel's exchange protocol and the matrix cores:
//   * the row sums go as {value, tag} words to one combiner workgroup per row and come back the same way (two hops);
//   * dHW_s (B x 4H: FOUR times the forward's h) goes to a slot per step, 16 columns x all rows per workgroup = whole lines,
//     write-through + flag, read by every workgroup with ordinary loads;
//   * dh_prev of the four owned units = dHW_s @ Wh[units, :]^T (K = 4H, N = 4) on v_mfma_f32_4x4x1_16b_f32: sixteen
//     independent 4 x 4 outer products per instruction = 64 batch rows x 4 units x one k (lane l supplies dHW[row l][k] and
//     Wh[unit l & 3][k]; result register r of lane l = row 4 (l / 4) + r, unit l & 3 -- tests/tools/micro/mfma4x4.hip); wave
//     (rg, kq) takes rows 64 rg ... and a k slice, the slices meet in LDS in a fixed order.  The owned rows of Wh stay in LDS
//     in the exchanged column order (k' = 16 * owner workgroup + 4 * gate + unit).
struct MidBwd {
    const float *d_out, *dhn, *dcn;            // (S,B,H) / (B,H) / (B,H); each may be null (= zero)
    const float *gates, *c, *c0, *xw, *hw, *stats, *gamma, *wh;
    float *dgate, *dxw, *dhw, *dh0, *dc0;
    float* dx;                                 // [S][nwg][4 gates][rows][4 units]: dHW_s as the workgroups exchange it
    u64* part_t;                               // [B][nwg][4] {value, tag}: a workgroup's part of the four row sums
    u64* fin_t;                                // [nrep][256][4] {value, tag}: the four row sums / 4H, from the row's combiner
    unsigned* flag;                            // [nrep][256]: steps whose dHW the workgroup has published
    int S, B, H, nwg, rgp /* 64-row groups (a power of two) */, ks /* k slices; rgp * ks <= 16 */, nrep;
    u64* prof;
};

__global__ __launch_bounds__(1024, 4) void lstm_mid_bwd_kernel(MidBwd a) {
    extern __shared__ float smem[];
    const int H = a.H, G = 4 * H, B = a.B, nwg = a.nwg, rgp = a.rgp, ks = a.ks;
    const int Bp = 64 * rgp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* Wt = smem;                          // [4 units][4H] in exchanged column order
    float* gp = Wt + (size_t)4 * G;            // [16 waves][64 rows][4 units]: k-slice partials of dh_prev
    float* rtot = gp + 4096;                   // [rows][4]: the row sums / 4H
    float* cst = rtot + 4 * kMidMaxB;          // [2][16]: gamma_x, gamma_h of the owned columns (n = gate * 4 + unit)
    float* cmb = cst + 32;                     // [4 * nwg]: a row's partial sums while its combiner adds them
    const int wg = blockIdx.x, j0 = wg * 4;
    for (int e = tid; e < 4 * G; e += 1024) {
        const int u = e / G, kk = e - u * G, n = kk & 15, w2 = kk >> 4;
        Wt[e] = a.wh[(size_t)(j0 + u) * G + (n >> 2) * H + 4 * w2 + (n & 3)];
    }
    if (tid < 16) {
        const int col = (tid >> 2) * H + j0 + (tid & 3);
        cst[tid] = a.gamma[col];
        cst[16 + tid] = a.gamma[G + col];
    }
    const int row = tid >> 2, jj = tid & 3, cj = j0 + jj;
    const bool cell = row < B;
    float dh_carry = 0.f, dc_carry = 0.f;
    if (cell) {
        if (a.dhn) dh_carry = a.dhn[(size_t)row * H + cj];
        if (a.dcn) dc_carry = a.dcn[(size_t)row * H + cj];
    }
    // product role: wave (rg, kq): rows 64 rg + lane, owner workgroups [kq * nb, (kq + 1) * nb) of the exchanged columns
    const int rg = wave & (rgp - 1), kq = wave / rgp;
    const bool active = kq < ks;
    const int nb = nwg / ks;
    int ar = 64 * rg + lane;
    if (ar >= B) ar = B - 1;                   // rows past the batch repeat the last one (results unused)
    const size_t slot = (size_t)nwg * Bp * 16;
    const size_t a_off = (size_t)(active ? kq : 0) * nb * Bp * 16 + (size_t)ar * 4;
    const float* bp = Wt + (size_t)(lane & 3) * G + (size_t)(active ? kq : 0) * nb * 16;
    const float inv_g = 1.f / (float)G;
    float sg[4], sx[4], sh[4], sst[4], sc_new = 0.f, sc_prev = 0.f, sdo = 0.f;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) sg[gg] = sx[gg] = sh[gg] = sst[gg] = 0.f;
    auto load_saved = [&](int s) {
        if (!cell) return;
        const size_t r = (size_t)s * B + row;
        quad_load4(a.gates + r * G + jj * H + j0, sg, lane);
        quad_load4(a.xw + r * G + jj * H + j0, sx, lane);
        quad_load4(a.hw + r * G + jj * H + j0, sh, lane);
        const vfloat4 st = *reinterpret_cast<const vfloat4*>(a.stats + r * 4);
        sst[0] = st.x; sst[1] = st.y; sst[2] = st.z; sst[3] = st.w;
        sc_new = a.c[r * H + cj];
        sc_prev = s == 0 ? a.c0[(size_t)row * H + cj] : a.c[(r - B) * H + cj];
        sdo = a.d_out ? a.d_out[r * H + cj] : 0.f;
    };
    if (a.S > 0) load_saved(a.S - 1);
    __syncthreads();

    u64 tprev_ = a.prof ? wall_clock64() : 0;
    for (int s = a.S - 1; s >= 0; --s) {
        const int t = a.S - 1 - s;                 // steps done before this one
        const uint32_t tag = (uint32_t)t + 1u;
        // ---- gate adjoints of the owned units, their part of the four LayerNorm-adjoint row sums -> the row's combiner
        float da[4] = {0.f, 0.f, 0.f, 0.f}, xh[4] = {0.f, 0.f, 0.f, 0.f}, hh[4] = {0.f, 0.f, 0.f, 0.f};
        float r4[4] = {0.f, 0.f, 0.f, 0.f};
        if (cell) {
            const float ig = sg[0], fg = sg[1], og = sg[2], ug = sg[3];
            const float dh = sdo + dh_carry;
            const float tc = tanhf(sc_new);
            const float dc = dc_carry + dh * og * (1.f - tc * tc);
            da[0] = dc * ug * ig * (1.f - ig);
            da[1] = dc * sc_prev * fg * (1.f - fg);
            da[2] = dh * tc * og * (1.f - og);
            da[3] = dc * ig * (1.f - ug * ug);
            dc_carry = dc * fg;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                xh[gg] = (sx[gg] - sst[0]) * sst[1];
                hh[gg] = (sh[gg] - sst[2]) * sst[3];
                const float dyx = da[gg] * cst[gg * 4 + jj], dyh = da[gg] * cst[16 + gg * 4 + jj];
                r4[0] += dyx; r4[1] += dyx * xh[gg];
                r4[2] += dyh; r4[3] += dyh * hh[gg];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) r4[q] = quad_sum(r4[q]);
        if (cell) xchg_put(a.part_t + ((size_t)row * nwg + wg) * 4 + jj, jj == 0 ? r4[0] : jj == 1 ? r4[1] : jj == 2 ? r4[2] : r4[3], tag);
        HPC_RLL_TICK(0)
        for (int r = wg; r < B; r += nwg) {        // rows this workgroup combines
            if (tid < 4 * nwg) {
                const u64* src = a.part_t + (size_t)r * nwg * 4 + tid;
                long spins = 0;
                u64 w;
                while ((uint32_t)((w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != tag)
                    mid_poll_failed(spins);
                cmb[tid] = __uint_as_float((uint32_t)w);
            }
            __syncthreads();
            if (tid < 256) {                        // wave q adds sum q over the workgroups, in workgroup order per lane
                const int q = tid >> 6;
                float tsum = 0.f;
                for (int w = lane; w < nwg; w += 64) tsum += cmb[w * 4 + q];
                tsum = wave_sum(tsum) * inv_g;
                if (lane < a.nrep) xchg_put(a.fin_t + (size_t)lane * 1024 + 4 * r + q, tsum, tag);
            }
            __syncthreads();
        }
        HPC_RLL_TICK(1)
        if (tid < 4 * B) {
            long spins = 0;
            u64 w;
            const u64* src = a.fin_t + (size_t)(wg % a.nrep) * 1024 + tid;
            while ((uint32_t)((w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != tag)
                mid_poll_failed(spins);
            rtot[tid] = __uint_as_float((uint32_t)w);
        }
        __syncthreads();
        HPC_RLL_TICK(2)
        // ---- dXW, dHW of the owned columns; dHW to the step's slot
        float o_dhw[4] = {0.f, 0.f, 0.f, 0.f}, o_dxw[4] = {0.f, 0.f, 0.f, 0.f};
        if (cell) {
            const vfloat4 rt = *reinterpret_cast<const vfloat4*>(rtot + 4 * row);
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const float dyx = da[gg] * cst[gg * 4 + jj], dyh = da[gg] * cst[16 + gg * 4 + jj];
                o_dhw[gg] = sst[3] * (dyh - rt.z - hh[gg] * rt.w);
                o_dxw[gg] = sst[1] * (dyx - rt.x - xh[gg] * rt.y);
                gp[row * 16 + gg * 4 + jj] = o_dhw[gg];      // (gp is free between two products)
            }
        }
        __syncthreads();
        HPC_RLL_TICK(6)
        if (cell) {   // the row's 16 values as four 16-byte pieces, ONE 16-byte write-through store per thread: a wave writes whole lines (4- or 8-byte pieces: 6.8-9 us until acknowledged)
            // slot layout [gate][row][unit]: a reader's load instruction (lane = row) then covers 64 x 16 contiguous bytes
            // (rows 64 bytes apart, the first layout, cost the L1 a tag lookup per lane: 19 us for the 1 MB of B = 64, H = 1024)
            const vfloat4 v = *reinterpret_cast<const vfloat4*>(gp + row * 16 + 4 * jj);
            mid_store4(a.dx + (size_t)t * slot + (size_t)wg * Bp * 16, (unsigned)((jj * Bp + row) * 16), v);
        }
        mid_publish(a.flag, a.nrep, tag);
        HPC_RLL_TICK(7)
        if (cell) {   // saved for the weight-gradient products: nobody in this launch waits for these
            const size_t r = (size_t)s * B + row;
            const size_t o = r * G + jj * H + j0;
            quad_store4(a.dhw + o, o_dhw, lane);
            quad_store4(a.dxw + o, o_dxw, lane);
            quad_store4(a.dgate + o, da, lane);
        }
        if (s > 0) load_saved(s - 1);              // (in flight during the product)
        HPC_RLL_TICK(3)
        mid_wait(a.flag + (wg % a.nrep) * 256, nwg, tag);
        HPC_RLL_TICK(4)
        // ---- dh_prev of the owned units = dHW_s @ Wh[units, :]^T
        if (active) {
            vfloat4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float* ap = a.dx + (size_t)t * slot + a_off;
            // chunks of two owner workgroups (eight 16-byte loads per lane), software-pipelined over two register buffers: the
            // loads of chunk c + 1 are in flight while chunk c is multiplied (every workgroup reads ALL of dHW_s: B * 16 KB at
            // H = 1024 -- the phase is bound by that transfer, not by the matrix instructions)
            vfloat4 b0v[8], b1v[8];
            auto issue = [&](vfloat4 (&v)[8], int c) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int bi = 2 * c + (u >> 2) < nb ? 2 * c + (u >> 2) : 0;
                    v[u] = *reinterpret_cast<const vfloat4*>(ap + ((size_t)bi * 4 + (u & 3)) * Bp * 4);
                }
            };
            auto mult = [&](const vfloat4 (&v)[8], int c) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (2 * c + (u >> 2) < nb) {
                        const vfloat4 bv = *reinterpret_cast<const vfloat4*>(bp + (size_t)(2 * c + (u >> 2)) * 16 + 4 * (u & 3));
                        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(v[u].x, bv.x, acc, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(v[u].y, bv.y, acc1, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(v[u].z, bv.z, acc, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(v[u].w, bv.w, acc1, 0, 0, 0);
                    }
            };
            // (every workgroup starts its walk over the chunks somewhere else: all of them reading the same 8 KB at the same
            // time puts 32 CUs of an XCD on one L2 channel at a time)
            const int nc = (nb + 1) / 2;
            const int rot = wg % nc;
            auto chunk = [&](int c) __attribute__((always_inline)) { const int x = c + rot; return x >= nc ? x - nc : x; };
            issue(b0v, chunk(0));
            for (int c = 0; c < nc; c += 2) {
                if (c + 1 < nc) issue(b1v, chunk(c + 1));
                mult(b0v, chunk(c));
                if (c + 2 < nc) issue(b0v, chunk(c + 2));
                if (c + 1 < nc) mult(b1v, chunk(c + 1));
            }
            acc += acc1;
            float* pw = gp + ((size_t)wave * 64 + 4 * (lane >> 2)) * 4 + (lane & 3);   // register r: row 4 (lane / 4) + r, unit lane & 3
#pragma unroll
            for (int r = 0; r < 4; ++r) pw[r * 4] = acc[r];
        }
        __syncthreads();
        if (cell) {
            const float* pr = gp + ((size_t)(row >> 6) * 64 + (row & 63)) * 4 + jj;
            float d = 0.f;
            for (int q = 0; q < ks; ++q) d += pr[(size_t)q * rgp * 256];
            dh_carry = d;
        }
        HPC_RLL_TICK(5)
    }
    if (cell) {
        a.dh0[(size_t)row * H + cj] = dh_carry;
        a.dc0[(size_t)row * H + cj] = dc_carry;
    }
}

struct MidBwdCfg { int rgp, ks, nwg; size_t lds; };
inline MidBwdCfg mid_bwd_cfg(int B, int H) {
    MidBwdCfg c;
    const int rgn = (B + 63) / 64;
    c.rgp = 1;
    while (c.rgp < rgn) c.rgp *= 2;
    c.nwg = H / 4;
    c.ks = 16 / c.rgp;
    while (c.ks > 1 && c.nwg % c.ks) c.ks /= 2;
    c.lds = ((size_t)16 * H + 4096 + 4 * kMidMaxB + 32 + 1024) * sizeof(float);
    return c;
}
// workspace of the backward: [flags: 32 x 256 words][fin_t: 32 x 256 x 4 u64][part_t: B * nwg * 4 u64] ... line-aligned [dx: a slot per step]
constexpr size_t kMidBwdFinFloats = (size_t)kMidMaxRep * 256 * 4 * 2;
inline size_t mid_bwd_ws_floats(int S, int B, int H) {
    if (!lstm_mid_shape(B, H) || S <= 0) return 0;
    const MidBwdCfg c = mid_bwd_cfg(B, H);
    return kMidFlagFloats + kMidBwdFinFloats + (size_t)B * c.nwg * 8 + 32 + (size_t)S * c.nwg * 64 * c.rgp * 16;
}
// Measured (tests/tools/r04_lstm_mid_table.py with HPC_RLL_TUNE=33:0 against 33:2, profiles/r04_lstm_mid_bwd.txt): every
// workgroup reads ALL of dHW_s (B x 4H: four times the forward's h) -- 14.8 us of the step at B = 64, H = 1024, where the
// step kernels' whole recurrence takes ~20.  The kernel wins where launches, not bytes, are the step: B <= 32 (whole backward
// per step, eager: 20.1 -> 17.4 us at B = 16, H = 384; 31.0 -> 28.9 at H = 1024; a tie as hipGraph replays) and loses above
// (B = 64: 36.9 -> 46.9).  Key 33 = 2 takes it for every mid-batch shape (tests, experiments).
inline bool mid_bwd_pays(int B, int H) { return g_lstm_mid_bwd >= 2 ? mid_pays(B, H) : B <= 32; }
inline bool mid_bwd_ok(int B, int H, hipStream_t st) {
    if (!g_lstm_mid_bwd || !g_lstm_mid || !g_lstm_persist || !lstm_mid_shape(B, H) || !mid_bwd_pays(B, H) || !persist_runtime_ready(st)) return false;
    const MidBwdCfg c = mid_bwd_cfg(B, H);
    return persist_resident_t(lstm_mid_bwd_kernel, 1024, c.nwg, c.lds);
}
inline int launch_mid_bwd(MidBwd a, float* ws_mid, int layer, hipStream_t st) {
    const MidBwdCfg c = mid_bwd_cfg(a.B, a.H);
    a.nwg = c.nwg;
    a.rgp = c.rgp;
    a.ks = c.ks;
    a.nrep = g_lstm_mid_rep < 1 ? 1 : g_lstm_mid_rep > kMidMaxRep ? kMidMaxRep : g_lstm_mid_rep;
    const size_t polled = kMidFlagFloats + kMidBwdFinFloats + (size_t)a.B * c.nwg * 8;
    a.flag = reinterpret_cast<unsigned*>(ws_mid);
    a.fin_t = reinterpret_cast<u64*>(ws_mid + kMidFlagFloats);
    a.part_t = a.fin_t + kMidBwdFinFloats / 2;
    a.dx = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws_mid + polled) + 127) & ~(uintptr_t)127);
    if (hipMemsetAsync(ws_mid, 0, polled * sizeof(float), st) != hipSuccess) return last_error();   // flags and tags
    if (c.lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void*)lstm_mid_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds);
        if (e != hipSuccess) return (int)e;
    }
    a.prof = persist_prof();
    persist_chain_before(st);
    hipLaunchKernelGGL(lstm_mid_bwd_kernel, dim3(c.nwg), dim3(1024), c.lds, st, a);
    persist_chain_after(st);
    persist_prof_report("mid-batch bwd: cell+partials combine(own rows) wait_sums saved_stores+prefetch_issue wait_dHW product finalize(to LDS) slot_store+ack+flag", layer, a.S, st);
    return last_error();
}

}  // namespace
}  // namespace hpc_rll
