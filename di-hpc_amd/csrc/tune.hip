// tune.hip -- hpc_rll_tune_set: the path switches of the library, ONE table (round 5).
//
// Rounds 1-4 grew 40 process-global knobs spread over the translation units, most of them launch parameters of measured-and-
// settled sweeps (now constants next to the code they tune) or switches to measured losers (removed, with the numbers left
// in a comment where the code was and in HISTORY.md).  What is left are switches between paths that BOTH ship -- each wins on
// some shapes under the default rule -- so that tests and measurement tools can force either one and compare them on the
// same inputs, plus the one deployment switch (key 3).  Keys keep their historical numbers; retired keys answer
// HPC_RLL_EINVAL.  Process-global, not synchronised: set them before launching work, from one thread (tests do).
#include "hpc_rll_hip.h"

namespace hpc_rll {
extern int g_lstm_persist, g_lstm_wave, g_gemm_tile256, g_scatter_lds_fwd, g_scatter_npb, g_scan_fold, g_split_algo, g_sample_batch,
    g_gemm_dma, g_lstm_block, g_lstm_block_skew, g_pad_wave, g_lstm_mid, g_onehot_fill_mb, g_ppo_fused, g_lstm_mid_bwd, g_onehot_qpw,
    g_scatter_build, g_scatter_bwd_xcd, g_scatter_bwd_tile;
namespace {
struct TuneKey {
    int key;
    int* var;
    int lo, hi;          // accepted range (inclusive)
    int allowed_mask;    // != 0: value & ~mask must be 0 (bit-mask keys)
    const char* doc;
};
const TuneKey kKeys[] = {
    {3, &g_lstm_persist, 0, 1, 0, "persistent (co-residency dependent) LSTM kernels: 1 on, 0 = step kernels only (GPUs shared between processes; HPC_RLL_LSTM_PERSIST=0)"},
    {8, &g_lstm_wave, 0, 1, 0, "B <= 4, L >= 2: 1 = all layers in one launch as a wavefront, 0 = one persistent kernel per layer"},
    {16, &g_gemm_tile256, 0, 1, 0, "256x256x16 tiles (16 waves) for interior products that fill the chip in whole rounds; 0 = 128-class tiles"},
    {17, &g_scatter_lds_fwd, 0, 1, 0, "ScatterConnection forward: 1 = LDS-staged streaming kernel where it applies, 0 = cells-per-thread kernel"},
    {18, &g_scatter_npb, 0, 64, 0, "channels per workgroup of the LDS-staged scatter kernel (a multiple of 4; 0 = by LDS budget)"},
    {21, &g_scan_fold, 0, 1, 0, "loss finalisation folded into the last workgroup of the launch (grids up to 512); 0 = separate finalize launch"},
    {22, &g_split_algo, 0, 1, 0, "group split of the padding ops: 0 = runs of equal keys, 1 = the element-level DP (host; small n)"},
    {24, &g_sample_batch, 0, 64, 0, "samples per wave of the large-batch C51 / QR-DQN forwards: 0 by batch size, 1 off, 8 / 16 / 32 / 64"},
    {25, &g_gemm_dma, 0, 2, 0, "LDS-DMA staged GEMM tiles: 1 all forms, 2 the 256x256 NT tile only, 0 register staging everywhere"},
    {26, &g_lstm_block, 0, 255, 1 | 8 | 128, "large-batch LSTM row-block kernels: bit 0 forward, bit 3 backward, bit 7 fence-free forward exchanges (default 9)"},
    {27, &g_lstm_block_skew, 0, 200, 0, "microseconds between the starts of consecutive row blocks of those kernels (default 10)"},
    {28, &g_pad_wave, 0, 1, 0, "packed Pad1D: 1 = wave tiles in output space, 0 = the workgroup kernel (identical results)"},
    {29, &g_lstm_mid, 0, 2, 0, "mid-batch persistent LSTM forward (5 <= B <= 256): 0 off, 1 one stream, 2 two streams where they fit"},
    {31, &g_onehot_fill_mb, 0, 65536, 0, "one-hot gradients of at least this many MiB are written as fill + values (0 = never)"},
    {32, &g_ppo_fused, 0, 1, 0, "PPO forward in one launch (1) or three (0)"},
    {33, &g_lstm_mid_bwd, 0, 2, 0, "mid-batch persistent LSTM backward: 0 off, 1 where it pays (B <= 32), 2 every mid-batch shape"},
    {35, &g_onehot_qpw, 0, 8192, 0, "16-byte quads per workgroup of the one-launch one-hot kernel (0 = by size, else 256 ... 8192)"},
    {37, &g_scatter_build, 0, 1, 0, "scatter owner table / chain links built inside the forward kernel (1) or by the index launch (0)"},
    {38, &g_scatter_bwd_xcd, 0, 2, 0, "scatter backward: XCD-major workgroup order where its pieces are below a 64-byte sector pair (1), always (2), or launch order (0)"},
    {40, &g_scatter_bwd_tile, 0, 2, 0, "scatter backward by spatial tiles (full entity rows from one writer): 0 = by rule (16-byte pieces and M*16 >= H*W), 1 = never (plane kernel), 2 = wherever it applies"},
};
}  // namespace
}  // namespace hpc_rll

extern "C" int hpc_rll_tune_set(int key, int value) {
    using namespace hpc_rll;
    for (const TuneKey& k : kKeys) {
        if (k.key != key) continue;
        if (value < k.lo || value > k.hi || (k.allowed_mask && (value & ~k.allowed_mask))) return HPC_RLL_EINVAL;
        if (key == 18 && (value % 4) != 0) return HPC_RLL_EINVAL;
        if (key == 24 && !(value == 0 || value == 1 || value == 8 || value == 16 || value == 32 || value == 64)) return HPC_RLL_EINVAL;
        if (key == 35 && value != 0 && value < 256) return HPC_RLL_EINVAL;
        *k.var = value;
        return HPC_RLL_OK;
    }
    return HPC_RLL_EINVAL;
}

// Number of live keys and their descriptions (tests/test_abi.py checks the count stays <= 20).
extern "C" int hpc_rll_tune_count(void) { return (int)(sizeof(hpc_rll::kKeys) / sizeof(hpc_rll::kKeys[0])); }
extern "C" const char* hpc_rll_tune_doc(int index, int* key) {
    const int n = hpc_rll_tune_count();
    if (index < 0 || index >= n) return nullptr;
    if (key) *key = hpc_rll::kKeys[index].key;
    return hpc_rll::kKeys[index].doc;
}
