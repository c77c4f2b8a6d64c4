// pad_group.hpp -- the group-split policy on RUNS of equal keys, shared by the host entry point
// (hpc_rll_oracle_split_group, pad_scatter.hip) and the device plan kernel (pad_group.hip).
//
// Reference: hpc_rll/origin/padding.py:11-50 (oracle_split_group: DP over ELEMENTS, O(group * n^2), exactly `group`
// non-empty groups, cost of a group = its largest key x its element count, ties resolved to the smallest split point)
// and src/rl_utils/padding.cu:44-108 (the same DP with stack VLAs).
//
// The list is sorted by key, so it is D runs of equal keys: val[0] < val[1] < ... , E[r] = number of elements in the
// first r runs (E[0] = 0, E[D] = n).  With M = min(group, n) groups:
//   * D >= M.  Every optimal solution cuts only at run ends and its M group maxima are distinct: a cut strictly inside
//     a run either has a larger maximum on its right (moving the cut to the run's end is strictly cheaper) or the same
//     maximum on both sides, and then one of the >= M distinct keys is no group's maximum -- merging the two equal
//     groups and cutting behind that key's run is strictly cheaper.  The element DP's smallest-split-point tie rule
//     therefore only ever chooses among run ends, and states whose prefix holds fewer distinct keys than groups are
//     never minimisers: the DP over the D+1 run boundaries (same recurrence, same tie rule) gives the SAME cuts.
//   * D < M.  Zero padding is reachable (every group inside one run) and the tie rule decides everything: from state
//     (i elements, j groups) the element DP picks k = max(s, j-1), s = start of the last run of the prefix -- the whole
//     last run if that leaves enough elements for the remaining groups, else singletons.  Closed form, no table.
// Both are O(M * D^2) / O(M) after an O(n) (host) or histogram (device) pass, against O(M n^2) in the reference and
// O(M n log n) for the divide-and-conquer DP of round 2; tests/test_host_logic.py checks all three against each other
// and against the goldens recorded from the reference.
#pragma once
#include <stdint.h>

namespace hpc_rll {

constexpr int64_t kSplitInf = INT64_MAX / 4;

// Backtracking for D < M (closed form).  pos[0..M] element positions, gmax[0..M-1] group maxima.  Single thread.
template <class I64, class I32>
__host__ __device__ inline void split_runs_few(const I64* val, const I64* E, int D, int64_t n, int M, I64* pos, I32* gmax) {
    int64_t i = n;
    int r = D;                                   // the prefix [0,i) ends inside (or at the end of) run r-1
    pos[M] = n;
    for (int j = M; j >= 1; --j) {
        while (r > 1 && E[r - 1] >= i) --r;      // run r-1 is the last run that has elements below i
        const int64_t s = E[r - 1];              // start of the last run of the prefix
        const int64_t k = s > (int64_t)(j - 1) ? s : (int64_t)(j - 1);
        gmax[j - 1] = (I32)val[r - 1];
        pos[j - 1] = k;
        i = k;
    }
}

// One DP state: best predecessor run boundary r' in [lo, r-1] for f_prev[r'] + val[r-1] * (E[r] - E[r']).
template <class I64>
__host__ __device__ inline void split_runs_state(const I64* val, const I64* E, const int64_t* fprev, int r, int lo,
                                                 int64_t* best_out, int32_t* arg_out) {
    int64_t best = kSplitInf;
    int32_t arg = lo;
    const int64_t v = val[r - 1], er = E[r];
    for (int q = lo; q < r; ++q) {
        const int64_t fp = fprev[q];
        if (fp >= kSplitInf) continue;
        const int64_t c = fp + v * (er - E[q]);
        if (c < best) { best = c; arg = q; }     // strict: the smallest boundary wins ties, like the reference
    }
    *best_out = best;
    *arg_out = arg;
}

}  // namespace hpc_rll
