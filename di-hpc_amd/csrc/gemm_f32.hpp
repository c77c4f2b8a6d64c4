// gemm_f32.hpp -- exact-fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32) for the LSTM gate GEMMs.
//
// Replaces the reference's cublasSgemm calls (src/torch_utils/network/lstm.cu:121-123,145-147,352-370).
// The f32-input MFMA is bit-for-bit an fmaf chain in k order (no TF32-style truncation exists on gfx950), so the
// results are plain fp32 matmul results at the 157 TFLOP/s matrix rate.
//
//   C (M x N, row stride ldc) (+)= A (M x K) * B (K x N)
// with arbitrary element strides for A and B, so the three layouts the LSTM needs are one kernel family:
//   NN  x @ W          A(m,k) = A[m*lda + k]   B(k,n) = B[k*ldb + n]
//   NT  dY @ W^T       A(m,k) = A[m*lda + k]   B(k,n) = W[n*ldb + k]
//   TN  X^T @ dY       A(m,k) = X[k*lda + m]   B(k,n) = B[k*ldb + n]
//
// Tiling: workgroup = 4 waves, block tile BM x BN x BK (BK = 16 or 32); each wave owns WM x WN MFMA blocks of 32 x 32.
//
// LDS layout (round 2).  An operand tile is stored ROW-major -- one row per m (A) / n (B), BK floats, no padding -- with
// the k axis PERMUTED inside the row: [k = 0,2,4,..,BK-2 | k = 1,3,..,BK-1].  Lane (x = lane & 31, h = lane >> 5) of the
// 32x32x2 MFMA consumes k = 2s + h at step s, i.e. exactly the h-th half of its row, in order: one ds_read_b128 feeds
// FOUR consecutive MFMA steps (the first version fetched one ds_read_b32 per operand and step from k-major tiles; an
// ablation build with nothing but those reads and the MFMAs stopped at 136 TFLOP/s = 87 % of peak,
// profiles/r02_gemm_ablate_before.txt -- every instruction a wave issues between two of its MFMAs delays the next one).
// The arithmetic order along k is unchanged, so results are bit-identical to the first version.  The 16-byte chunks of
// a row are XOR-swizzled with the row index ((x>>2)&3 for 64-byte rows, (x>>1)&7 for 128-byte rows), which makes the
// b128 operand reads of 16 consecutive rows hit 16 different bank quads; the swizzle of rows x and x+32 is the same, so
// a wave's WM (WN) blocks share one address register and differ by an immediate offset.
// Staging per operand (register prefetch of the next k-tile + LDS double buffer, one barrier per k-tile):
//   * contiguous along k   : float4 global loads along k; the even / odd k's of each float4 go out as two 8-byte writes;
//   * contiguous along m/n : four float4 loads from four k-rows of equal parity (k, k+2, k+4, k+6), a 4x4 register
//     transpose, four 16-byte writes (one per row);
//   * anything else        : guarded scalar loads (odd sizes / unaligned pointers).
// When both tiles together are 256 4x4 blocks or fewer, A is staged by the first waves and B by the others (one block
// per thread); otherwise every thread stages its share of both.
#pragma once
#include <hip/hip_runtime.h>

#include "wave.hpp"

namespace hpc_rll {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float gf4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K;
    long a_sm, a_sk;   // A(m,k) = A[m*a_sm + k*a_sk]
    long b_sk, b_sn;   // B(k,n) = B[k*b_sk + n*b_sn]
    long ldc;
    int accumulate;    // C += A*B instead of C = A*B
    int splitk = 1;    // > 1: gridDim.z slices of K (gemm_splitk), slice z writes its PARTIAL product to
    long c_split = 0;  //      C + z*c_split; the consumer adds the slices in a fixed order (deterministic)
    int big = 0;       // 1: a long-K product that brings its own split-K (gemm_splitk_big): 128x128 tiles
    int xcd_swizzle = 0;   // set by launch_gemm_tile
    float* rowpart = nullptr;   // gemm_f32_nn_dma_kernel<., true>: per-row (mean, M2) of every 128-column block of C,
                                // [N/128][M][2] -- LayerNorm partials out of the product's epilogue (lstm_block.hpp)
};

enum GemmMode { kContigMN = 0, kContigK = 1, kGeneric = 2, kDmaK = 3 };   // kDmaK: k-contiguous, staged by LDS-DMA (DmaStage)

// LDS address (in floats) of position pp (0..BK-1) of the PERMUTED row x of a tile with BK floats per row.
template <int BK> __device__ __forceinline__ int lds_sw(int x) { return BK == 16 ? ((x >> 2) & 3) : BK == 32 ? ((x >> 1) & 7) : (x & 15); }   // (64: 256-byte rows, 16 chunks)
template <int BK> __device__ __forceinline__ int lds_pos(int x, int pp) {
    return x * BK + ((((pp >> 2) ^ lds_sw<BK>(x)) << 2) | (pp & 3));
}
// position of k inside the permuted row: evens first, then odds
template <int BK> __device__ __forceinline__ int kperm(int k) { return (k & 1) * (BK / 2) + (k >> 1); }

typedef float gf2 __attribute__((ext_vector_type(2)));

// Round 3: a k-contiguous operand staged by LDS-DMA (global_load_lds_dwordx4, gfx950): no staging registers, no ds_write,
// and -- what the ablation of round 2 had isolated as the cost of the HBM/L2 stream -- no register-returning vector-memory
// instruction between the MFMAs (tests/tools/micro/gemm_ablate.hip ABL 4, profiles/r03_gemm_ablate_dma.txt: 4096^3
// 132.7 -> 138-139 TFLOP/s, the C4 x-branch shape 130.9 -> 136-137).  A DMA piece lands as 64 lanes x 16 bytes CONTIGUOUS at
// a wave-uniform LDS base, so the tile's layout is dictated: 16-byte slot e of the tile <-> (row e / CPR, physical chunk
// e % CPR).  What stays free is WHICH 16 bytes of global memory a lane asks for, and that is enough for the XOR swizzle of
// lds_pos(): the lane of slot (row, pc) fetches the row's logical chunk pc ^ lds_sw(row).  What is NOT possible at 16-byte
// granularity is the even/odd k permutation inside a row, so rows are kept RAW (k in memory order) and the permutation
// moves into the meaning of the MFMA steps: lane half h still reads the h-th half of its row with ds_read_b128, but that
// half is now k = 8h .. 8h+7 of the k-tile (BK = 16), i.e. MFMA step s multiplies k = s and k = 8 + s.  Both operands must
// use the convention (kDmaK x kDmaK only); each output is still one exact fp32 fma chain, in another k order than the
// register-staged kernels (same order for every tile shape that takes this path).
template <int X, int BK, int NT>
struct DmaStage {
    static constexpr int CPR = BK / 4;                   // 16-byte chunks per row
    static constexpr int NP = X * CPR / NT;              // pieces per thread and k-tile
    static_assert((X * CPR) % NT == 0 && NT % 64 == 0, "DMA staging thread set");
    const float* p[NP];
    __device__ __forceinline__ void init(const float* __restrict__ base, long sx, int x0, int k0) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int e = j * NT + (int)threadIdx.x, r = e / CPR, pc = e % CPR;
            p[j] = base + (long)(x0 + r) * sx + k0 + 4 * (pc ^ lds_sw<BK>(r));
        }
    }
    __device__ __forceinline__ void issue(float* tile) {     // one k-tile: NP pieces, then step to the next k-tile
        typedef __attribute__((address_space(3))) void* lds_ptr;
        typedef const __attribute__((address_space(1))) void* gl_ptr;
        const int wave = (int)threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            __builtin_amdgcn_global_load_lds((gl_ptr)p[j], (lds_ptr)(tile + (j * NT + wave * 64) * 4), 16, 0, 0);
            p[j] += BK;
        }
    }
};

// Stages one operand tile (X rows-of-the-operand by BK k) through registers, using threads [T0, T0+NT) of the workgroup.
//   MODE kContigMN: element (x,k) at base[x + k*sk]        (unit stride along x)
//   MODE kContigK : element (x,k) at base[x*sx + k]        (unit stride along k)
//   MODE kGeneric : element (x,k) at base[x*sx + k*sk], fully guarded
// Work unit of the two vector modes: a 4 x 4 block = four float4 loads.
//   kContigK : block e -> rows xr = 4*(e / KQ) .. +3, k = 4*(e % KQ) .. +3            (KQ = BK/4)
//   kContigMN: block e -> x = 4*(e / NSET) .. +3, k-set c = e % NSET = four k of equal parity
//              {par + 2*(4*idx + t), t = 0..3}, par = c / (BK/8), idx = c % (BK/8): positions 4c .. 4c+3 of the row.
template <int X, int BK, int MODE, int T0, int NT>
struct TileStage {
    static constexpr int KQ = BK / 4;                    // float4 per row of the tile (k-contiguous operand)
    static constexpr int NSET = BK / 4;                  // k-sets (x-contiguous operand)
    static constexpr int NBLK = X * BK / 16;             // 4x4 blocks in the tile
    static constexpr int NB = (NBLK + NT - 1) / NT;      // blocks per staging thread
    static constexpr int NS = X * BK / NT;               // scalars per staging thread (generic)
    static_assert(NT % 64 == 0 && T0 % 64 == 0 && NT % KQ == 0 && (X * BK) % NT == 0, "staging thread set");
    gf4 v[MODE == kGeneric ? 1 : NB * 4];
    float s[MODE == kGeneric ? NS : 1];
    const float* p;   // this thread's first element of the NEXT k-tile to prefetch (fast path)

    static __device__ __forceinline__ bool active() { return (int)threadIdx.x >= T0 && (int)threadIdx.x < T0 + NT; }
    static __device__ __forceinline__ int kset_first(int c) { return c / (BK / 8) + 8 * (c % (BK / 8)); }

    // Fast path for tiles that lie entirely inside the operand (uniform per workgroup and k-tile): no per-lane
    // bounds test, so no divergent branch around a load.  With the guarded form the compiler had to assume 0..NV
    // loads in flight at every merge point and protected the zero-fill of the next operand's registers with
    // s_waitcnt vmcnt(0) -- the A prefetch was waited for BEFORE the MFMAs of the current k-tile, i.e. the register
    // prefetch hid nothing of the A latency.  Addresses are one per-thread pointer plus uniform (scalar) offsets,
    // advanced once per k-tile (the guarded form recomputed 64-bit products per load).
    __device__ __forceinline__ void init(const float* __restrict__ base, long sx, long sk, int x0, int k0) {
        const int lt = (int)threadIdx.x - T0;
        if (MODE == kContigMN) p = base + (long)(k0 + kset_first(lt % NSET)) * sk + x0 + 4 * (lt / NSET);
        else if (MODE == kContigK) p = base + (long)(x0 + 4 * (lt / KQ)) * sx + k0 + (lt % KQ) * 4;
        else p = base;
    }
    __device__ __forceinline__ void advance(long sk) { p += (MODE == kContigMN) ? BK * sk : BK; }
    __device__ __forceinline__ void load_fast(long sx, long sk) {
        if (!active()) return;
        const int lt = (int)threadIdx.x - T0;
        if (MODE == kContigMN) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (lt + i * NT >= NBLK) break;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    v[i * 4 + t] = *reinterpret_cast<const gf4*>(p + (long)(2 * t) * sk + i * 4 * (NT / NSET));
            }
        } else if (MODE == kContigK) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (lt + i * NT >= NBLK) break;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[i * 4 + r] = *reinterpret_cast<const gf4*>(p + (long)(i * 4 * (NT / KQ) + r) * sx);
            }
        }
    }

    // ABL 4 (tests/tools/micro/gemm_ablate.hip only, WRONG results): the loads of load_fast as LDS-DMA pieces
    // (global_load_lds_dwordx4: 64 lanes x 16 bytes land contiguously at an M0 base, no VGPRs, no ds_write) -- what the
    // staging would cost if the tile layout were the DMA's.  The LDS positions are arbitrary inside the tile.
    template <int SZ = 16>   // SZ = 4: the same bytes as 4-byte pieces (any layout / transposition possible, 4x the instructions)
    __device__ __forceinline__ void dma_fast(long sx, long sk, float* lds_tile) {
        if (!active()) return;
        const int lt = (int)threadIdx.x - T0;
        typedef __attribute__((address_space(3))) void* lds_ptr;
        typedef const __attribute__((address_space(1))) void* gl_ptr;
        const int wv = lt >> 6;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (lt + i * NT >= NBLK) break;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float* src = MODE == kContigMN ? p + (long)(2 * t) * sk + i * 4 * (NT / NSET)
                                                     : p + (long)(i * 4 * (NT / KQ) + t) * sx;
                if constexpr (SZ == 16) {
                    float* dst = lds_tile + (((i * 4 + t) * (NT / 64) + wv) * 256) % (X * BK);
                    __builtin_amdgcn_global_load_lds((gl_ptr)src, (lds_ptr)dst, 16, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float* dst = lds_tile + ((((i * 4 + t) * 4 + e) * (NT / 64) + wv) * 64) % (X * BK);
                        __builtin_amdgcn_global_load_lds((gl_ptr)(src + e), (lds_ptr)dst, 4, 0, 0);
                    }
                }
            }
        }
    }

    __device__ __forceinline__ void load(const float* __restrict__ base, long sx, long sk, int x0, int k0, int XD,
                                         int KD) {
        if (!active()) return;
        const int lt = (int)threadIdx.x - T0;
        if (MODE == kContigMN) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int e = lt + i * NT;
                if (e >= NBLK) break;
                const int x = x0 + 4 * (e / NSET), kf = k0 + kset_first(e % NSET);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    gf4 w = {0.f, 0.f, 0.f, 0.f};
                    if (kf + 2 * t < KD && x < XD) w = *reinterpret_cast<const gf4*>(base + (long)(kf + 2 * t) * sk + x);
                    v[i * 4 + t] = w;
                }
            }
        } else if (MODE == kContigK) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int e = lt + i * NT;
                if (e >= NBLK) break;
                const int xr = x0 + 4 * (e / KQ), k = k0 + (e % KQ) * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int x = xr + r;
                    x = x < XD ? x : XD - 1;   // clamp: rows beyond the edge are never stored to C
                    gf4 w = {0.f, 0.f, 0.f, 0.f};
                    if (k < KD) w = *reinterpret_cast<const gf4*>(base + (long)x * sx + k);
                    v[i * 4 + r] = w;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int e = lt + i * NT;
                const int k = k0 + e / X, x = x0 + e % X;
                s[i] = (k < KD && x < XD) ? base[(long)x * sx + (long)k * sk] : 0.f;
            }
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ tile) const {
        if (!active()) return;
        const int lt = (int)threadIdx.x - T0;
        if (MODE == kContigMN) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int e = lt + i * NT;
                if (e >= NBLK) break;
                const int c = e % NSET, x = 4 * (e / NSET);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const gf4 w = {v[i * 4 + 0][j], v[i * 4 + 1][j], v[i * 4 + 2][j], v[i * 4 + 3][j]};
                    *reinterpret_cast<gf4*>(tile + lds_pos<BK>(x + j, 4 * c)) = w;
                }
            }
        } else if (MODE == kContigK) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int e = lt + i * NT;
                if (e >= NBLK) break;
                const int xr = 4 * (e / KQ), kh = (e % KQ) * 2;   // k = 2*kh .. 2*kh+3 -> even slots kh, kh+1 ; odd BK/2 + kh, +1
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const gf4 w = v[i * 4 + r];
                    const gf2 ev = {w[0], w[2]}, od = {w[1], w[3]};
                    *reinterpret_cast<gf2*>(tile + lds_pos<BK>(xr + r, kh)) = ev;
                    *reinterpret_cast<gf2*>(tile + lds_pos<BK>(xr + r, BK / 2 + kh)) = od;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int e = lt + i * NT;
                tile[lds_pos<BK>(e % X, kperm<BK>(e / X))] = s[i];
            }
        }
    }
};

// BM x BN x BK block tile, 256 threads.  WM x WN = MFMA blocks per wave; waves arranged (BM/(32*WM)) x (BN/(32*WN)).
// Resident workgroups per CU the LDS footprint allows (160 KB per CU); the register allocation is steered to match
// (launch_bounds' second argument = waves per SIMD): at 136 registers the 128x128x16 tile ran 3 workgroups per CU, so
// a 1024-tile product (the C4 recurrent GEMM) ran as 768 + a 256-workgroup tail at one workgroup per CU.
template <int BM, int BN, int BK, int NW = 4> struct GemmOcc {
    static constexpr int lds = 2 * BK * (BM + BN) * 4;
    // (256x128 tiles hold 128 accumulator registers per lane: two waves per SIMD; a 16-wave workgroup IS the CU's
    //  four waves per SIMD: one workgroup per CU)
    static constexpr int value = NW == 16 ? 1 : NW == 8 ? 4 : BM * BN >= 256 * 128 ? 2 : lds <= 36 * 1024 ? 4 : lds <= 53 * 1024 ? 3 : 2;   // NW == 8: two 8-wave workgroups per CU
};

// INTERIOR: every tile of the launch lies entirely inside A, B and its K slice (M % BM == N % BN == K % BK == 0, no
// generic operand) -- decided on the host, so that each instantiation holds ONE main loop: the branch-free prefetch
// loop (see TileStage::load_fast) or the guarded one.  (Both loops in one kernel behind a uniform branch cost 20-40
// registers and spilled in the NT variants.)
// ABL != 0: ablation builds for tests/tools/micro/gemm_ablate.hip only (WRONG results): 1 = MFMAs + LDS operand reads
// only; 2 = + global prefetch (waited for where the LDS store would be); 3 = + LDS store + barrier, no global loads;
// 4 = staging by LDS-DMA (global_load_lds_dwordx4) + barrier instead of register prefetch + ds_write; 5 = the same bytes
// as 4-byte LDS-DMA pieces (what an m/n-contiguous operand would need); 6 (with kDmaK operands; CORRECT results) = the
// LDS-DMA loop with three buffers and the barrier in the middle of a tile: 256x256 / 16 waves 140.0 -> 142.5 TFLOP/s, but
// the 8-wave tile loses its second workgroup per CU to the third buffer (143.2 -> 136.9), so the shipped loop has two.
// Where the time goes (tests/tools/micro/gemm_ablate.hip, profiles/r02_gemm_ablate.txt; 4096^3 NN, full-entropy data,
// shader clock measured in-run at 2.41 GHz): MFMAs + the b128 operand reads alone 150 TFLOP/s (95 % of the matrix
// rate; 136 with the first version's b32 reads); + LDS store and barrier 141; + the global prefetch 129 -- the same
// with one or TWO k-tiles of prefetch in flight (a second register set, tried and removed), with 128x128 or 256x128
// tiles, with full or half cache lines per load (TN / NT / NN all within 127-130): what the stream costs is not its
// latency but the ISSUE of the vector-memory instructions beside the MFMAs (~65 cycles of matrix pipe per
// global_load_dwordx4 of a wave; MI355X_MICROARCH.md quotes ~60 for an LDS-DMA piece -- measured in round 3 (ABL 4): a
// dwordx4 LDS-DMA piece costs nothing measurable, the kernel then runs at the rate of "LDS store + barrier, no global
// loads"; hence DmaStage below for the products whose operands are both k-contiguous).
// NW: waves per workgroup (4, or 16 for the 256x256 tile: half the vector-memory instructions per MFMA).
template <int BM, int BN, int BK, int WM, int WN, int AMODE, int BMODE, bool INTERIOR, int ABL = 0, int NW = 4>
__global__ __launch_bounds__(NW * 64, (GemmOcc<BM, BN, BK, NW>::value)) void gemm_f32_kernel(const GemmArgs g) {
    constexpr int WAVES_M = BM / (32 * WM);
    constexpr int NTH = NW * 64;
    static_assert(WAVES_M * (BN / (32 * WN)) == NW, "waves per workgroup");
    constexpr bool PIPE3 = (AMODE == kDmaK) && ABL == 6;   // three LDS buffers, barrier in the middle of a tile (see the DMA loop)
    constexpr int NBUF = PIPE3 ? 3 : 2;
    __shared__ __attribute__((aligned(16))) float lds[NBUF * BK * BM + NBUF * BK * BN];
    float* const As = lds;                    // [buf][BK*BM]
    float* const Bs = lds + NBUF * BK * BM;   // [buf][BK*BN]

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    // XCD-aware tile order.  Workgroups are dispatched round-robin over the 8 XCDs (id % 8), each with its own 4 MB L2.
    // Give XCD x the x-th contiguous eighth of the tile sequence, and walk that sequence in groups of 8 tile rows
    // (column-major inside a group), so that the ~128 workgroups resident on one XCD cover a compact 8 x 16 block of
    // tiles and share their A rows / B columns through that L2.
    int tile_m = blockIdx.y, tile_n = blockIdx.x;
    if (g.xcd_swizzle) {
        const int nbx = gridDim.x, nby = gridDim.y, total = nbx * nby;
        const int id = blockIdx.y * nbx + blockIdx.x;
        const int per = total >> 3;                       // host sets xcd_swizzle only when total % 8 == 0
        const int t = (id & 7) * per + (id >> 3);
        constexpr int GROUP_M = 8;
        const int in_group = GROUP_M * nbx;
        const int grp = t / in_group, first_m = grp * GROUP_M;
        const int gsz = (nby - first_m) < GROUP_M ? (nby - first_m) : GROUP_M;
        const int r = t - grp * in_group;
        tile_m = first_m + r % gsz;
        tile_n = r / gsz;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // who stages what: both tiles together <= 256 blocks -> disjoint wave sets, one block per thread
    constexpr int NBA = BM * BK / 16, NBB = BN * BK / 16;
    constexpr bool SPLIT = NBA + NBB <= NTH && NBA % 64 == 0 && NBB % 64 == 0;
    TileStage<BM, BK, AMODE == kDmaK ? kContigK : AMODE, 0, SPLIT ? NBA : NTH> sa;     // (unused with LDS-DMA staging)
    TileStage<BN, BK, BMODE == kDmaK ? kContigK : BMODE, SPLIT ? NBA : 0, SPLIT ? NBB : NTH> sb;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // split-K: slice z owns k-tiles [z*per, (z+1)*per)
    const int all_tiles = (g.K + BK - 1) / BK;
    const int per = (all_tiles + g.splitk - 1) / g.splitk;
    const int kbeg = (int)blockIdx.z * per * BK;
    const int KE = min(g.K, kbeg + per * BK);            // exclusive k end of this slice
    const int ktiles = KE > kbeg ? (KE - kbeg + BK - 1) / BK : 0;
    float* const Cz = g.C + (long)blockIdx.z * g.c_split;
    constexpr bool DMA = AMODE == kDmaK;
    static_assert((AMODE == kDmaK) == (BMODE == kDmaK), "LDS-DMA staging: both operands or none (k convention)");
    static_assert(!DMA || (INTERIOR && (BK == 16 || BK == 32) && (ABL == 0 || ABL == 6)), "LDS-DMA staging: interior tiles, 64- or 128-byte rows");
    static_assert(!PIPE3 || BK == 16, "pipelined DMA loop: two operand quarters per tile");
    DmaStage<BM, BK, NTH> da;
    DmaStage<BN, BK, NTH> db;
    if constexpr (DMA) {
        da.init(g.A, g.a_sm, m0, kbeg);
        db.init(g.B, g.b_sn, n0, kbeg);
        if (ktiles > 0) {
            da.issue(As);
            db.issue(Bs);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        sa.load(g.A, g.a_sm, g.a_sk, m0, kbeg, g.M, KE);
        sb.load(g.B, g.b_sn, g.b_sk, n0, kbeg, g.N, KE);
        sa.store(As);
        sb.store(Bs);
    }
    __syncthreads();
    // operand fetch: lane (x = lane & 31, h = lane >> 5) reads the h-th half of its permuted row, 16 bytes = 4 MFMA steps
    // at a time; block i of the wave sits 32 rows further (same swizzle): an immediate offset
    constexpr int NQ = BK / 8;   // 16-byte chunks per half row
    int a_off[NQ], b_off[NQ];
    {
        const int am = wm * 32 * WM + (lane & 31), bn = wn * 32 * WN + (lane & 31), hh = (lane >> 5) * (BK / 2);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            a_off[q] = lds_pos<BK>(am, hh + 4 * q);
            b_off[q] = lds_pos<BK>(bn, hh + 4 * q);
        }
    }
    // One k-tile of MFMAs out of LDS buffer `buf`.
    auto mfma_tile = [&](int buf) __attribute__((always_inline)) {
        const float* __restrict__ as = As + buf * BK * BM;
        const float* __restrict__ bs = Bs + buf * BK * BN;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            gf4 a[WM], b[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) a[i] = *reinterpret_cast<const gf4*>(as + a_off[q] + i * 32 * BK);
#pragma unroll
            for (int j = 0; j < WN; ++j) b[j] = *reinterpret_cast<const gf4*>(bs + b_off[q] + j * 32 * BK);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (PIPE3) {
        // Three LDS buffers, ONE barrier per k-tile placed between the two operand quarters of a tile: when a wave reaches
        // it, the operands of the second quarter are already in registers (no refill bubble behind the barrier), and the
        // first quarter of the NEXT tile -- visible to everyone as of this barrier -- is fetched while the second quarter's
        // MFMAs run, so the matrix pipe never waits for ds_read at a tile boundary.  After the barrier every wave has left
        // tile kt-1, whose buffer takes the request for tile kt+2.
        gf4 a0[WM], b0[WN], a1[WM], b1[WN];
        auto fetch = [&](int buf, int q, gf4 (&a)[WM], gf4 (&b)[WN]) __attribute__((always_inline)) {
            const float* __restrict__ as = As + buf * BK * BM;
            const float* __restrict__ bs = Bs + buf * BK * BN;
#pragma unroll
            for (int i = 0; i < WM; ++i) a[i] = *reinterpret_cast<const gf4*>(as + a_off[q] + i * 32 * BK);
#pragma unroll
            for (int j = 0; j < WN; ++j) b[j] = *reinterpret_cast<const gf4*>(bs + b_off[q] + j * 32 * BK);
        };
        auto mfma4 = [&](const gf4 (&a)[WM], const gf4 (&b)[WN]) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
        };
        if (ktiles > 1) {
            da.issue(As + 1 * BK * BM);
            db.issue(Bs + 1 * BK * BN);
        }
        if (ktiles > 0) fetch(0, 0, a0, b0);
        int cur = 0;
        for (int kt = 0; kt < ktiles; ++kt) {
            const int nxt = cur == 2 ? 0 : cur + 1, nn = nxt == 2 ? 0 : nxt + 1;
            fetch(cur, 1, a1, b1);
            mfma4(a0, b0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of tile kt+1
            __syncthreads();                                   // everyone's; and every wave is past tile kt-1
            if (kt + 2 < ktiles) {
                da.issue(As + nn * BK * BM);
                db.issue(Bs + nn * BK * BN);
            }
            if (kt + 1 < ktiles) fetch(nxt, 0, a0, b0);
            mfma4(a1, b1);
            cur = nxt;
        }
    } else if constexpr (DMA) {
        // next k-tile requested straight into the other LDS buffer (every wave has passed the barrier that ended the last
        // reads of that buffer), MFMAs of this one, then: my pieces have landed (vmcnt) + everyone's (barrier)
        for (int kt = 0; kt < ktiles; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < ktiles) {
                da.issue(As + (buf ^ 1) * BK * BM);
                db.issue(Bs + (buf ^ 1) * BK * BN);
            }
            mfma_tile(buf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else if constexpr (INTERIOR) {
        sa.init(g.A, g.a_sm, g.a_sk, m0, kbeg + BK);
        sb.init(g.B, g.b_sn, g.b_sk, n0, kbeg + BK);
        for (int kt = 0; kt < ktiles; ++kt) {
            const int buf = kt & 1;
            if constexpr (ABL != 0) asm volatile("" ::: "memory");
            if (kt + 1 < ktiles && (ABL == 0 || ABL == 2)) {
                sa.load_fast(g.a_sm, g.a_sk);
                sb.load_fast(g.b_sn, g.b_sk);
                sa.advance(g.a_sk);
                sb.advance(g.b_sk);
            }
            if constexpr (ABL == 4 || ABL == 5) {
                if (kt + 1 < ktiles) {
                    sa.template dma_fast<ABL == 4 ? 16 : 4>(g.a_sm, g.a_sk, As + (buf ^ 1) * BK * BM);
                    sb.template dma_fast<ABL == 4 ? 16 : 4>(g.b_sn, g.b_sk, Bs + (buf ^ 1) * BK * BN);
                    sa.advance(g.a_sk);
                    sb.advance(g.b_sk);
                }
            }
            mfma_tile(buf);
            if constexpr (ABL == 4 || ABL == 5) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            if (kt + 1 < ktiles && (ABL == 0 || ABL == 3)) {
                sa.store(As + (buf ^ 1) * BK * BM);
                sb.store(Bs + (buf ^ 1) * BK * BN);
            }
            if constexpr (ABL == 2) {   // the prefetched registers must have arrived here, as for the LDS store
#pragma unroll
                for (int i = 0; i < (int)(sizeof(sa.v) / sizeof(sa.v[0])); ++i) asm volatile("" : : "v"(sa.v[i]));
#pragma unroll
                for (int i = 0; i < (int)(sizeof(sb.v) / sizeof(sb.v[0])); ++i) asm volatile("" : : "v"(sb.v[i]));
            }
            if constexpr (ABL == 0 || ABL == 3) __syncthreads();
        }
    } else {
        for (int kt = 0; kt < ktiles; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < ktiles) {
                sa.load(g.A, g.a_sm, g.a_sk, m0, kbeg + (kt + 1) * BK, g.M, KE);
                sb.load(g.B, g.b_sn, g.b_sk, n0, kbeg + (kt + 1) * BK, g.N, KE);
            }
            mfma_tile(buf);
            if (kt + 1 < ktiles) {
                sa.store(As + (buf ^ 1) * BK * BM);
                sb.store(Bs + (buf ^ 1) * BK * BN);
            }
            __syncthreads();
        }
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0 + wn * 32 * WN + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < g.M && n < g.N) {
                    float* p = Cz + (long)m * g.ldc + n;
                    *p = g.accumulate ? (*p + acc[i][j][r]) : acc[i][j][r];
                }
            }
        }
}

// ---- TN products (A and B both contiguous along m / n, rows along k: the LSTM's weight gradients, K = S*B) by LDS-DMA.
// A 16-byte piece of such an operand is four consecutive m at ONE k, so the tile can only land k-MAJOR in LDS: [BK][BM].
// A 32x32x2 MFMA wants, per lane (i = lane & 31, h = lane >> 5), A[m_i][k_h] -- with a k-major tile the 32 lanes of a half
// read 32 consecutive floats, which is one ds_read_b32 per MFMA step and block: the instruction rate round 1 stopped at
// (136 TFLOP/s for the bare inner loop).  The way out is to let one 16-byte read serve FOUR m-blocks: block j of the wave
// owns the rows m = 4 i + j (interleaved, not contiguous), so lane i's float4 at [k][4 i .. 4 i + 3] is exactly its operand
// for blocks 0..3 at that k; B likewise with a float2 and two n-blocks (n = 2 i + j).  A wave is then 4 x 2 blocks =
// 128 (m) x 64 (n), 128 accumulator registers, and step t of a k-tile needs ONE ds_read_b128 + ONE ds_read_b64 for
// 8 MFMAs (the NT kernel: one b128 per 4).  k is consumed in memory order, two per MFMA (k = 2t from the lower half wave,
// 2t + 1 from the upper): the same order as the register-staged kernels, so results are bit-identical to theirs.
// Workgroup = 8 waves (2 along m x 4 along n) = a 256 x 256 tile, one per CU (2 waves per SIMD at ~150 registers); a
// wave-piece of the DMA is one whole k-row of a tile (64 lanes x 16 B = 256 floats), global reads of 1 KiB.
// Interior launches only (M % 256 == N % 256 == K % 16 == 0, 16-byte aligned rows), split-K slices as in gemm_f32_kernel.
__global__ __launch_bounds__(512, 2) void gemm_f32_tn_dma_kernel(const GemmArgs g) {
    constexpr int BM = 256, BN = 256, BK = 16, NW = 8;
    __shared__ __attribute__((aligned(16))) float lds[2 * BK * BM + 2 * BK * BN];
    float* const As = lds;                 // [buf][BK][BM]
    float* const Bs = lds + 2 * BK * BM;   // [buf][BK][BN]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave & 1, wn = wave >> 1;                    // 2 x 4 waves
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int all_tiles = g.K / BK;
    const int per = (all_tiles + g.splitk - 1) / g.splitk;
    const int kbeg = (int)blockIdx.z * per * BK;
    const int KE = min(g.K, kbeg + per * BK);
    const int ktiles = KE > kbeg ? (KE - kbeg) / BK : 0;
    float* const Cz = g.C + (long)blockIdx.z * g.c_split;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // DMA: wave w brings k-rows w and w + 8 of both tiles; lane L the 16 bytes at m0 + 4 L of that row
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gl_ptr;
    const float* pa = g.A + (long)(kbeg + wave) * g.a_sk + m0 + 4 * lane;
    const float* pb = g.B + (long)(kbeg + wave) * g.b_sk + n0 + 4 * lane;
    const long a8 = 8 * g.a_sk, b8 = 8 * g.b_sk, a16 = 16 * g.a_sk, b16 = 16 * g.b_sk;
    auto issue = [&](int buf) __attribute__((always_inline)) {
        float* at = As + buf * BK * BM + wave * BM;
        float* bt = Bs + buf * BK * BN + wave * BN;
        __builtin_amdgcn_global_load_lds((gl_ptr)pa, (lds_ptr)at, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gl_ptr)(pa + a8), (lds_ptr)(at + 8 * BM), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gl_ptr)pb, (lds_ptr)bt, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gl_ptr)(pb + b8), (lds_ptr)(bt + 8 * BN), 16, 0, 0);
        pa += a16;
        pb += b16;
    };
    if (ktiles > 0) issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int h = lane >> 5, i32 = lane & 31;
    const int a_off = h * BM + wm * 128 + 4 * i32;             // + 2 t BM: row k = 2 t + h
    const int b_off = h * BN + wn * 64 + 2 * i32;
    for (int kt = 0; kt < ktiles; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < ktiles) issue(buf ^ 1);
        const float* __restrict__ as = As + buf * BK * BM + a_off;
        const float* __restrict__ bs = Bs + buf * BK * BN + b_off;
#pragma unroll
        for (int t = 0; t < BK / 2; ++t) {
            const gf4 a = *reinterpret_cast<const gf4*>(as + 2 * t * BM);
            const gf2 b = *reinterpret_cast<const gf2*>(bs + 2 * t * BN);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // epilogue: block (i, j) of the wave holds rows m = 4 * row32 + i, columns n = 2 * col32 + j (interleaved); the 32x32
    // C layout: col32 = lane & 31, row32 = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).  The two n-blocks of a row pair up into
    // one 8-byte store: 32 lanes x 8 B = 256 contiguous bytes per row.
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row32 = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int m = m0 + wm * 128 + 4 * row32 + i;
            float* p = Cz + (long)m * g.ldc + n0 + wn * 64 + 2 * i32;
            gf2 o = {acc[i][0][r], acc[i][1][r]};
            if (g.accumulate) { o[0] += p[0]; o[1] += p[1]; }
            *reinterpret_cast<gf2*>(p) = o;
        }
}

// ---- NN products (A contiguous along k, B along n) by LDS-DMA: the A tile as in the NT kernels (raw rows, DmaStage), the
// B tile k-major as in gemm_f32_tn_dma_kernel.  The k convention of the raw A rows (lane half h holds k = 8h .. 8h+7, MFMA
// step s multiplies k = s and k = 8 + s) is met on the B side by reading ROW 8h + s of the k-major tile; n-blocks are
// interleaved by four (n = 4 i + j), so one ds_read_b128 of B feeds the four n-blocks of a step and one ds_read_b128 of A
// feeds four steps of an m-block: a wave of 2 x 4 blocks = 64 (m) x 128 (n) issues 6 reads per 32 MFMAs.  Workgroup = 8
// waves (4 along m x 2 along n) = 256 x 256, one per CU.  Same k order as the NT DMA kernels: bit-identical to them on the
// same operands.
template <bool PIPE, bool ROWSTATS = false>   // PIPE: three LDS buffers, the k-tile barrier between the two halves of a tile (see the loop): measured
                       // 142.9 -> 144.0 TFLOP/s at 4096^3, 141.5 -> 141.9 on the C4 x-branch shape -- shipped: false
                       // ROWSTATS: also write g.rowpart (single K slice, no accumulate)
__global__ __launch_bounds__(512, 2) void gemm_f32_nn_dma_kernel(const GemmArgs g) {
    constexpr int BM = 256, BN = 256, BK = 16, NTH = 512, NBUF = PIPE ? 3 : 2;
    __shared__ __attribute__((aligned(16))) float lds[NBUF * BK * BM + NBUF * BK * BN];
    float* const As = lds;                    // [buf][BM rows][BK]  (swizzled chunks, lds_pos)
    float* const Bs = lds + NBUF * BK * BM;   // [buf][BK][BN]       (k-major)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave & 3, wn = wave >> 2;                    // 4 x 2 waves
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int all_tiles = g.K / BK;
    const int per = (all_tiles + g.splitk - 1) / g.splitk;
    const int kbeg = (int)blockIdx.z * per * BK;
    const int KE = min(g.K, kbeg + per * BK);
    const int ktiles = KE > kbeg ? (KE - kbeg) / BK : 0;
    float* const Cz = g.C + (long)blockIdx.z * g.c_split;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gl_ptr;
    DmaStage<BM, BK, NTH> da;
    da.init(g.A, g.a_sm, m0, kbeg);
    const float* pb = g.B + (long)(kbeg + wave) * g.b_sk + n0 + 4 * lane;     // wave w: k-rows w and w + 8
    const long b8 = 8 * g.b_sk, b16 = 16 * g.b_sk;
    auto issue = [&](int buf) __attribute__((always_inline)) {
        da.issue(As + buf * BK * BM);
        float* bt = Bs + buf * BK * BN + wave * BN;
        __builtin_amdgcn_global_load_lds((gl_ptr)pb, (lds_ptr)bt, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gl_ptr)(pb + b8), (lds_ptr)(bt + 8 * BN), 16, 0, 0);
        pb += b16;
    };
    if (ktiles > 0) issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int h = lane >> 5, i32 = lane & 31;
    int a_off[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) a_off[q] = lds_pos<BK>(wm * 64 + i32, 8 * h + 4 * q);   // block i: + 32 rows = + 32 * BK floats
    const int b_off = (8 * h) * BN + wn * 128 + 4 * i32;                                 // + s * BN: row 8 h + s
    auto half = [&](const float* as, const float* bs, int q, const gf4 (&a)[2], gf4 b0) __attribute__((always_inline)) {
        // steps 4 q .. 4 q + 3 of a tile: A operands and the first B operand already in registers
        gf4 b = b0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            gf4 bnx = b;
            if (t < 3) bnx = *reinterpret_cast<const gf4*>(bs + (4 * q + t + 1) * BN);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j], acc[i][j], 0, 0, 0);
            b = bnx;
        }
    };
    if constexpr (PIPE) {
        // one barrier per k-tile, BETWEEN the two halves of a tile: a wave reaches it with the second half's first operands
        // in registers, and fetches the next tile's first operands -- visible as of this barrier -- under the second half's
        // MFMAs: no operand refill bubble behind a barrier.  After it every wave has left tile kt-1, whose buffer takes the
        // request for tile kt+2 (three buffers).
        if (ktiles > 1) issue(1);
        gf4 a0[2], a1[2], b0, b1;
        if (ktiles > 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a0[i] = *reinterpret_cast<const gf4*>(As + a_off[0] + i * 32 * BK);
            b0 = *reinterpret_cast<const gf4*>(Bs + b_off);
        }
        int cur = 0;
        for (int kt = 0; kt < ktiles; ++kt) {
            const int nxt = cur == 2 ? 0 : cur + 1, nn = nxt == 2 ? 0 : nxt + 1;
            const float* as = As + cur * BK * BM;
            const float* bs = Bs + cur * BK * BN + b_off;
#pragma unroll
            for (int i = 0; i < 2; ++i) a1[i] = *reinterpret_cast<const gf4*>(as + a_off[1] + i * 32 * BK);
            b1 = *reinterpret_cast<const gf4*>(bs + 4 * BN);
            half(as, bs, 0, a0, b0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 2 < ktiles) issue(nn);
            if (kt + 1 < ktiles) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a0[i] = *reinterpret_cast<const gf4*>(As + nxt * BK * BM + a_off[0] + i * 32 * BK);
                b0 = *reinterpret_cast<const gf4*>(Bs + nxt * BK * BN + b_off);
            }
            half(as, bs, 1, a1, b1);
            cur = nxt;
        }
    } else {
        for (int kt = 0; kt < ktiles; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < ktiles) issue(buf ^ 1);
            const float* __restrict__ as = As + buf * BK * BM;
            const float* __restrict__ bs = Bs + buf * BK * BN + b_off;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                gf4 a[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const gf4*>(as + a_off[q] + i * 32 * BK);
                half(as, bs, q, a, *reinterpret_cast<const gf4*>(bs + 4 * q * BN));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    // epilogue: block (i, j): rows m = 32 i + row32 (plain), columns n = 4 col32 + j (interleaved): the four n-blocks of a
    // row are one 16-byte store, 32 lanes x 16 B = 512 contiguous bytes per row
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row32 = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int m = m0 + wm * 64 + i * 32 + row32;
            float* p = Cz + (long)m * g.ldc + n0 + wn * 128 + 4 * i32;
            gf4 o = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
            if (g.accumulate) o += *reinterpret_cast<const gf4*>(p);
            *reinterpret_cast<gf4*>(p) = o;
        }
    if constexpr (ROWSTATS) {
        // a wave holds 64 rows x 128 columns: per row, the 32 lanes of a half wave sum their four columns -- mean, then M2
        // around that mean (two passes over registers); lane i32 = 16 i + r keeps row (i, r) and writes it
        float my_m = 0.f, my_d = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sm = half_sum_all((acc[i][0][r] + acc[i][1][r]) + (acc[i][2][r] + acc[i][3][r]));
                const float m = sm * (1.f / 128.f);
                float d = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) d += (acc[i][j][r] - m) * (acc[i][j][r] - m);
                d = half_sum_all(d);
                if (i32 == 16 * i + r) { my_m = m; my_d = d; }
            }
        const int rr = i32 & 15;
        const long m = m0 + wm * 64 + (i32 >> 4) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
        const int cb = blockIdx.x * 2 + wn;
        *reinterpret_cast<gf2*>(g.rowpart + ((size_t)cb * g.M + m) * 2) = gf2{my_m, my_d};
    }
}

inline bool gemm_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Would launch_gemm take the 256x256x16 tile for this problem (and, with both operands k-contiguous, its LDS-DMA form)?
// Callers that can present an operand either way (lstm.hip: a weight or its transposed copy) ask before choosing.
inline bool gemm_tile256_ok(int M, int N, int K, int splitk) {
    extern int g_gemm_tile256;
    const int sk = splitk > 1 ? splitk : 1;
    const long wgs = (long)(M / 256) * (N / 256) * sk;
    return g_gemm_tile256 && M > 0 && N > 0 && M % 256 == 0 && N % 256 == 0 && K % 16 == 0 && (wgs % 256 == 0 || wgs >= 4096);
}
// LDS-DMA tiles for NT products (tune key 25): 1 = 256x128x16 with 8 waves, TWO workgroups per CU -- while one of them
// sits at its k-tile barrier the other keeps the matrix pipe fed (142.7 TFLOP/s at 4096^3 against 140.0 for the 16-wave
// 256x256 tile, whose workgroup IS the CU; profiles/r03_gemm_ablate_dma.txt) -- when the workgroups come in whole rounds
// of two per CU; 2 = 256x256x16 with 16 waves under the conditions of gemm_tile256_ok; 0 = neither.
inline int gemm_dma_tile(int M, int N, int K, int splitk) {
    extern int g_gemm_dma;   // 0 off, 1 both tiles, 2 the 256x256 tile only
    if (!g_gemm_dma || M <= 0 || N <= 0 || K % 16 != 0) return 0;
    const int sk = splitk > 1 ? splitk : 1;
    if (g_gemm_dma == 1 && M % 256 == 0 && N % 128 == 0) {
        const long wgs = (long)(M / 256) * (N / 128) * sk;
        if (wgs % 512 == 0 || wgs >= 8192) return 1;
    }
    return gemm_tile256_ok(M, N, K, splitk) ? 2 : 0;
}
inline bool gemm_dma_ok(int M, int N, int K, int splitk) { return gemm_dma_tile(M, N, K, splitk) != 0; }
// ... and would an NN product (A along k, B along n, 16-byte aligned C rows) take gemm_f32_nn_dma_kernel?
inline bool gemm_nn_dma_ok(int M, int N, int K, int splitk) {
    extern int g_gemm_dma;
    const int sk = splitk > 1 ? splitk : 1;
    const long wgs = (long)(M / 256) * (N / 256) * sk;
    return g_gemm_dma == 1 && M > 0 && N > 0 && M % 256 == 0 && N % 256 == 0 && K % 16 == 0 && (wgs % 256 == 0 || wgs >= 4096);
}

// Staging mode an operand admits.  x = the operand's non-k axis (m for A, n for B).
inline int gemm_mode(const float* p, long sx, long sk, int XD, int KD) {
    if (sx == 1 && (sk % 4) == 0 && (XD % 4) == 0 && gemm_al16(p)) return kContigMN;
    if (sk == 1 && (sx % 4) == 0 && (KD % 4) == 0 && gemm_al16(p)) return kContigK;
    return kGeneric;
}

template <int BM, int BN, int BK, int WM, int WN>
inline void launch_gemm_tile(const GemmArgs& g_in, int am, int bm, hipStream_t st) {
    const dim3 grid((g_in.N + BN - 1) / BN, (g_in.M + BM - 1) / BM, g_in.splitk > 1 ? g_in.splitk : 1);
    GemmArgs g = g_in;
    g.xcd_swizzle = (((long)grid.x * grid.y) % 8 == 0 && grid.y >= 8) ? 1 : 0;
    const bool interior = g.M % BM == 0 && g.N % BN == 0 && g.K % BK == 0;
#define HPC_RLL_GEMM_CASE(AM, BMD)                                                                          \
    if (am == AM && bm == BMD) {                                                                            \
        if (interior)                                                                                       \
            hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, WM, WN, AM, BMD, true>), grid, dim3(256), 0, st, g); \
        else                                                                                                \
            hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, WM, WN, AM, BMD, false>), grid, dim3(256), 0, st, g); \
        return;                                                                                             \
    }
    HPC_RLL_GEMM_CASE(kContigK, kContigMN)    // NN
    HPC_RLL_GEMM_CASE(kContigK, kContigK)     // NT
    HPC_RLL_GEMM_CASE(kContigMN, kContigMN)   // TN
#undef HPC_RLL_GEMM_CASE
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, WM, WN, kGeneric, kGeneric, false>), grid, dim3(256), 0, st, g);
}

// Tile shape launch_gemm picks for an (M, N) problem.
struct GemmTile { int bm, bn; };
inline GemmTile gemm_tile_of(int M, int N) {
    // Enough workgroups to keep >= 2 resident per CU (256 CUs): a lone 128x128 workgroup per CU cannot overlap its
    // own staging with its MFMAs.
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    const long t64 = (long)((M + 127) / 128) * ((N + 63) / 64);
    if (M <= 32) return {32, 128};   // skinny: the 4 waves side by side along N
    if (t128 >= 512) return {128, 128};
    if (t64 >= 512 || M > 64) return {128, 64};
    return {64, 64};
}

// Number of K slices worth using when the output tiles alone cannot fill the 256 CUs (the per-step recurrent
// products of the LSTM at B <~ 1024: a handful of workgroups walking a long K serially is latency bound -- measured
// 98 us per step for dHW(64x2048) @ Wh^T on 8 workgroups).  Slices write partial products; the consumer sums them in
// a fixed order.  Every slice keeps >= 2 k-tiles of 32.
inline int gemm_splitk(int M, int N, int K) {
    const GemmTile t = gemm_tile_of(M, N);
    const long tiles = (long)((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn);
    const int ktiles = (K + 31) / 32;
    int s = 1;
    constexpr int g_gemm_lat_target = 256;   // one workgroup per CU
    // latency regime: fill the CUs (in-process sweep over mid-batch LSTM shapes: 256 workgroups up to M = 256, two per CU
    // from M = 512: B=1024,H=512 4.62/7.37 -> 4.11/7.02 ms)
    const long lat_target = M >= 512 ? 2L * g_gemm_lat_target : g_gemm_lat_target;
    while (s < 16 && tiles * s < lat_target && ktiles / (s * 2) >= 2) s *= 2;
    constexpr int g_gemm_thr_ktiles = 8;   // (in-process sweep, B = 512..2048: 8 -> forward -3..6 %, backward +-1 %; C4 unaffected)
    while (s < 16 && tiles * s < 768 && ktiles / (s * 2) >= g_gemm_thr_ktiles) s *= 2;   // throughput regime: 3-4 workgroups per CU
    return s;                                                             // while the slices stay long (C4 dh: 2)
}

// The same for the large once-per-layer products with a long K (the weight gradients, K = S*B): fill the chip with
// ~2 workgroups per CU but keep >= 16 k-tiles per slice; the partial products are summed by a reduction kernel.
inline GemmTile gemm_tile_big(int M, int N, int K) {
    constexpr int g_gemm_big_target = 768;   // workgroups the split-K of the weight-gradient products aims for
    if (M > 64 && N > 64) {   // 128x128 only if its split-K can still reach the target
        const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
        const int ktiles = (K + 31) / 32;
        int s = 1;
        while (s < 16 && ktiles / (s * 2) >= 16) s *= 2;
        if (tiles * s >= g_gemm_big_target) return {128, 128};
    }
    return gemm_tile_of(M, N);
}
inline int gemm_splitk_big(int M, int N, int K) {
    const GemmTile t = gemm_tile_big(M, N, K);
    const long tiles = (long)((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn);
    const int ktiles = (K + 31) / 32;
    constexpr int g_gemm_big_target = 768;
    int s = 1;
    while (s < 16 && tiles * s < g_gemm_big_target && ktiles / (s * 2) >= 16) s *= 2;
    return s;
}

// C = A B through gemm_f32_nn_dma_kernel with the LayerNorm row partials in the epilogue (g.rowpart).  The caller has checked
// gemm_nn_rowstats_ok.
inline bool gemm_nn_rowstats_ok(const GemmArgs& g) {
    extern int g_gemm_dma;
    return g_gemm_dma == 1 && g.M > 0 && g.M % 256 == 0 && g.N % 256 == 0 && g.K % 16 == 0 && g.K > 0 && g.a_sk == 1 &&
           g.b_sn == 1 && (g.a_sm % 4) == 0 && (g.b_sk % 4) == 0 && (g.ldc % 4) == 0 && gemm_al16(g.A) && gemm_al16(g.B) &&
           gemm_al16(g.C) && g.splitk <= 1 && !g.accumulate && g.rowpart && (reinterpret_cast<uintptr_t>(g.rowpart) & 7) == 0;
}
inline void launch_gemm_nn_rowstats(const GemmArgs& g, hipStream_t st) {
    hipLaunchKernelGGL((gemm_f32_nn_dma_kernel<false, true>), dim3(g.N / 256, g.M / 256, 1), dim3(512), 0, st, g);
}

inline void launch_gemm(const GemmArgs& g, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0) return;
    const int am = gemm_mode(g.A, g.a_sm, g.a_sk, g.M, g.K);
    const int bm = gemm_mode(g.B, g.b_sn, g.b_sk, g.N, g.K);
    // measured at the LSTM shapes: NN runs better with BK = 16 (4-5 workgroups resident per CU: 108 vs 94 TFLOP/s on
    // the recurrent GEMM), NT is indifferent; the long-K weight-gradient products (TN, `big`) run 128x128x16 tiles
    // with their own split-K (4 workgroups per CU in one round: C4 backward 179 -> 171 ms vs 128x64x32)
    const GemmTile t = g.big ? gemm_tile_big(g.M, g.N, g.K) : gemm_tile_of(g.M, g.N);
    const int bk = ((am == kContigK && bm == kContigMN) || (g.big && t.bm == 128 && t.bn == 128)) ? 16 : 32;
    extern int g_gemm_tile256;   // tuning knob (hpc_rll_tune_set key 16)
    {   // TN: LDS-DMA with k-major tiles (gemm_f32_tn_dma_kernel), one 8-wave workgroup per CU
        extern int g_gemm_dma;
        const int sk = g.splitk > 1 ? g.splitk : 1;
        const long wgs = (long)(g.M / 256) * (g.N / 256) * sk;
        if (g_gemm_dma == 1 && am == kContigMN && bm == kContigMN && g.M % 256 == 0 && g.N % 256 == 0 && g.K % 16 == 0 &&
            (g.ldc % 2) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 7) == 0 && (g.c_split % 2) == 0 &&
            (wgs % 256 == 0 || wgs >= 4096)) {
            hipLaunchKernelGGL(gemm_f32_tn_dma_kernel, dim3(g.N / 256, g.M / 256, sk), dim3(512), 0, st, g);
            return;
        }
    }
    {   // NN: LDS-DMA, A as raw rows + B k-major (gemm_f32_nn_dma_kernel), one 8-wave workgroup per CU
        extern int g_gemm_dma;
        const int sk = g.splitk > 1 ? g.splitk : 1;
        const long wgs = (long)(g.M / 256) * (g.N / 256) * sk;
        if (g_gemm_dma == 1 && am == kContigK && bm == kContigMN && g.M % 256 == 0 && g.N % 256 == 0 && g.K % 16 == 0 &&
            (g.ldc % 4) == 0 && gemm_al16(g.C) && (g.c_split % 4) == 0 && (wgs % 256 == 0 || wgs >= 4096)) {
            hipLaunchKernelGGL(gemm_f32_nn_dma_kernel<false>, dim3(g.N / 256, g.M / 256, sk), dim3(512), 0, st, g);
            return;
        }
    }
    if (am == kContigK && bm == kContigK) {   // NT: LDS-DMA staged tiles (DmaStage)
        const int dt = gemm_dma_tile(g.M, g.N, g.K, g.splitk);
        const int sk = g.splitk > 1 ? g.splitk : 1;
        GemmArgs h = g;
        if (dt == 1) {
            const dim3 grid(g.N / 128, g.M / 256, sk);
            h.xcd_swizzle = (((long)grid.x * grid.y) % 8 == 0 && grid.y >= 8) ? 1 : 0;
            hipLaunchKernelGGL((gemm_f32_kernel<256, 128, 16, 2, 2, kDmaK, kDmaK, true, 0, 8>), grid, dim3(512), 0, st, h);
            return;
        }
        if (dt == 2) {
            const dim3 grid(g.N / 256, g.M / 256, sk);
            h.xcd_swizzle = 0;
            hipLaunchKernelGGL((gemm_f32_kernel<256, 256, 16, 2, 2, kDmaK, kDmaK, true, 0, 16>), grid, dim3(1024), 0, st, h);
            return;
        }
    }
    {
        // 256x256x16 tiles, 16 waves (one workgroup = a CU's four waves per SIMD): half the vector-memory instructions per
        // MFMA of the 128x128 tile -- the cost the ablation isolates (profiles/r02_gemm_ablate.txt: 4096^3 128.9 ->
        // 136.5 TFLOP/s, K=1024 x 65536 rows 128.4 -> 133.7; same k order, bit-identical results).  One workgroup per CU
        // means coarse rounds: only when the workgroup count is a multiple of the CU count or the tail is negligible.
        const int sk = g.splitk > 1 ? g.splitk : 1;
        const long wgs = (long)(g.M / 256) * (g.N / 256) * sk;
        if (g_gemm_tile256 && am != kGeneric && bm != kGeneric && g.M % 256 == 0 && g.N % 256 == 0 && g.K % 16 == 0 &&
            (wgs % 256 == 0 || wgs >= 4096)) {
            const dim3 grid(g.N / 256, g.M / 256, sk);
            GemmArgs h = g;
            h.xcd_swizzle = 0;   // measured neutral for this tile (136.5 vs 136.7)
#define HPC_RLL_GEMM256(AM, BMD)                                                                                          \
            if (am == AM && bm == BMD) {                                                                                  \
                hipLaunchKernelGGL((gemm_f32_kernel<256, 256, 16, 2, 2, AM, BMD, true, 0, 16>), grid, dim3(1024), 0, st, h); \
                return;                                                                                                   \
            }
            HPC_RLL_GEMM256(kContigK, kContigMN) HPC_RLL_GEMM256(kContigK, kContigK) HPC_RLL_GEMM256(kContigMN, kContigMN)
#undef HPC_RLL_GEMM256
        }
    }
    if (t.bm == 32) launch_gemm_tile<32, 128, 32, 1, 1>(g, am, bm, st);
    else if (t.bm == 128 && t.bn == 128) {
        if (bk == 16) launch_gemm_tile<128, 128, 16, 2, 2>(g, am, bm, st);
        else launch_gemm_tile<128, 128, 32, 2, 2>(g, am, bm, st);
    } else if (t.bm == 128) {
        if (bk == 16) launch_gemm_tile<128, 64, 16, 2, 1>(g, am, bm, st);
        else launch_gemm_tile<128, 64, 32, 2, 1>(g, am, bm, st);
    } else launch_gemm_tile<64, 64, 32, 1, 1>(g, am, bm, st);
}

}  // namespace hpc_rll
