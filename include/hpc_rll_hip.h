/*
 * hpc_rll_hip.h -- C ABI of libhpc_rll_hip.so, the MI355X (gfx950) operator library behind the
 * hpc_rll.rl_utils / hpc_rll.torch_utils Python API.
 *
 * This is the drop-in boundary: plain pointers + sizes + a HIP stream, no torch types, no
 * allocation, no host synchronisation, never throws.  Every entry point returns 0 on success,
 * a positive hipError_t value if the HIP runtime reported one, or a negative HPC_RLL_E* code for
 * an invalid argument.  All tensors are dense row-major ("contiguous") device buffers; floats are
 * fp32, indices int64 (torch.long), exactly as the reference's kernels assume.
 *
 * Each declaration cites the reference interface (file:line under /root/reference) it replaces:
 * the pybind entry `Fn(std::vector<Tensor> inputs, std::vector<Tensor>& outputs, scalars...)`
 * declared in include/hpc/rll/cuda/rl_utils/entry.h and .../torch_utils/network/entry.h.
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).
 * "scale" arguments let a data-parallel caller pass 1/(GLOBAL element count) so that per-rank
 * partial losses sum (one RCCL all-reduce) to the single-GPU result.
 */
#ifndef HPC_RLL_HIP_H_
#define HPC_RLL_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPC_RLL_OK 0
#define HPC_RLL_EINVAL (-1)     /* bad size / null pointer                      */
#define HPC_RLL_EALIGN (-2)     /* pointer not 4-byte aligned                   */
#define HPC_RLL_EUNSUPPORTED (-3) /* shape outside what the kernels implement   */
#define HPC_RLL_ETIMEOUT (-4)   /* a persistent LSTM kernel gave up waiting for its co-resident workgroups */

/* ABI version, bumped on any signature change or added entry point (3 = round 3: kernel timing, packed / grouped pad). */
int hpc_rll_abi_version(void);
/* Human readable message for a status returned by any entry point (static storage). */
const char* hpc_rll_status_string(int status);
/* Stream helpers for host bindings that must not call the HIP runtime themselves (the torch extension is host-only
 * C++): is_capturing returns 1 while `stream` records into a hipGraph, 0 if not, a negative code on error;
 * synchronize blocks the host until the stream has drained (positive hipError_t on failure). */
int hpc_rll_stream_is_capturing(void* stream);
int hpc_rll_stream_synchronize(void* stream);

/* ------------------------------------------------------------------------------------------
 * GAE  -- replaces GaeForward (include/hpc/rll/cuda/rl_utils/entry.h:62-66, src/rl_utils/gae.cu:8-28,
 *         kernel gae_kernel.h:10-29).  The reference has no backward (hpc_rll/rl_utils/gae.py:17-18);
 *         hpc_rll_gae_backward is the analytic adjoint of hpc_rll.origin.gae.
 *
 *   value (T+1,B), reward (T,B) -> adv (T,B)
 *   grad_adv (T,B) -> grad_value (T+1,B) [may be NULL], grad_reward (T,B) [may be NULL]
 *
 * `coef` is a T-float device table c_t = gamma*lambda*D_{t+1}/D_t (D_t = 1 + lambda*D_{t+1},
 * D_T = 0) that depends only on (T, gamma, lambda); fill it once with hpc_rll_gae_coef and reuse
 * it for every call with the same (T, gamma, lambda).
 * ------------------------------------------------------------------------------------------ */
int hpc_rll_gae_coef(float* coef, int T, float gamma, float lambda, void* stream);
int hpc_rll_gae_forward(const float* value, const float* reward, float* adv, const float* coef,
                        int T, int B, float gamma, void* stream);
int hpc_rll_gae_backward(const float* grad_adv, float* grad_value, float* grad_reward, const float* coef,
                         int T, int B, float gamma, void* stream);
/* Expert entry points: pin the launch configuration instead of the built-in heuristic.
 * vec in {1,2,4} columns per lane, lc in {2,4,8,16} time steps per wave chunk, nw in {1,2,4,8,16}
 * waves per workgroup; 0 for any of them = choose automatically.  flags: bit0 nontemporal loads,
 * bit1 nontemporal stores; -1 = choose automatically.  HPC_RLL_EUNSUPPORTED if the combination is
 * not instantiated. */
int hpc_rll_gae_forward_ex(const float* value, const float* reward, float* adv, const float* coef,
                           int T, int B, float gamma, int vec, int lc, int nw, int flags, void* stream);
int hpc_rll_gae_backward_ex(const float* grad_adv, float* grad_value, float* grad_reward, const float* coef,
                            int T, int B, float gamma, int vec, int lc, int nw, int flags, void* stream);
/* Flags bit 2 (value 4): half-wave tiles (32 columns, two time chunks per wave; narrow batches).  Bit 3 (value 8, with explicit
 * vec/lc/nw): the software-pipelined kernel (the next chunk's row loads are issued before the current chunk's barrier /
 * stores); bit-identical results.  Bits 4 and 5 (the one-trajectory-per-wavefront mapping and XCD-contiguous column tiles of
 * rounds 3-4) left the library in round 5 (tests/tools/micro/gae_wpt.hip keeps the former): HPC_RLL_EUNSUPPORTED.
 *
 * Diagnostics (no reference counterpart; the reference times whole python calls, tests/test_gae.py:31-52).
 * hpc_rll_ktime_begin(capacity) arms per-launch KERNEL timing for the next `capacity` GAE launches of this process
 * (start/stop events attached to the dispatch itself: the kernel's own begin/end, what rocprofv3 --kernel-trace
 * reports); hpc_rll_ktime_end waits for them, writes durations in milliseconds and kinds (0 forward, 1 backward) in
 * launch order, disarms and returns the number recorded.  Not thread-safe, not usable under stream capture.
 * hpc_rll_gae_last_config(dir, out[6]): the configuration the latest forward (dir 0) / backward (dir 1) launch used:
 * {columns per lane, steps per chunk, waves per workgroup, nontemporal flags, half-wave tiles, pipelined}. */
int hpc_rll_ktime_begin(int capacity);
int hpc_rll_ktime_end(float* ms, int* kind, int max);
int hpc_rll_gae_last_config(int dir, int* out);

/* ------------------------------------------------------------------------------------------
 * Shared helpers of the scalar-loss ops.
 *   hpc_rll_partials_floats(n): floats of scratch ("partials") an op over n columns/samples needs.
 *   hpc_rll_scale_rows: out[i] = g[0]*in[i] (i < n_in), 0 (n_in <= i < n_out) -- the generic
 *     "upstream scalar x saved unit gradient" backward step.
 * Every *_forward writes per-rank SUMS times `scale`; pass scale = 1/(global count) so that a
 * data-parallel caller gets the global mean with one all-reduce(sum) of the loss scalars.
 * ------------------------------------------------------------------------------------------ */
int64_t hpc_rll_partials_floats(int64_t n);
int hpc_rll_scale_rows(const float* g, const float* in, float* out, int64_t n_in, int64_t n_out, void* stream);

/* Categorical head over `rows` rows of N logits (rows = T*B or B), action int64 per row.
 * Replaces categoricalTarget/categoricalBehaviour (vtrace_kernel.h:11-151), crossEntropyKernel
 * (upgo_kernel.h:40-81), categoricalProbEntropy/categoricalProb (ppo_kernel.h:12-150) and the
 * backward kernels that consume their saved buffers.  forward: logp[row] = log softmax(logits)[a],
 * entropy[row] (nullable).  backward: grad[row,i] = g_logp*coef_logp[row]*(1[i==a]-p_i)
 *   + g_ent*coef_ent[row]*(-p_i*(log p_i + H));  g_* are device scalars (NULL = 1), coef_ent may
 *   be NULL (no entropy term). */
int hpc_rll_categorical_forward(const float* logits, const int64_t* action, float* logp, float* entropy,
                                int64_t rows, int N, void* stream);
int hpc_rll_categorical_backward(const float* logits, const int64_t* action, const float* coef_logp,
                                 const float* g_logp, const float* coef_ent, const float* g_ent,
                                 float* grad_logits, int64_t rows, int N, void* stream);

/* Path switches (a test / measurement hook, not a serving API; csrc/tune.hip holds the table): process-global plain ints,
 * not synchronised -- set them while no other thread is launching work, and never between an LSTM forward and its backward.
 * Every key selects between paths that BOTH ship (each wins on some shapes under the default rule), so that tests can run one
 * against the other on the same inputs; key 3 is also the deployment switch for GPUs shared between processes.  The launch
 * parameters and the measured-and-rejected variants that rounds 1-4 exposed as keys 0-2, 4-7, 9-15, 19, 20, 23, 30, 34, 36, 39
 * are constants / gone (HISTORY.md has what each measured); those keys answer HPC_RLL_EINVAL.
 *   key  3  persistent (co-residency dependent) LSTM kernels: 1 (default) / 0 = step kernels only
 *   key  8  B <= 4, L >= 2: 1 (default) = all layers in one launch as a wavefront / 0 = one persistent kernel per layer
 *   key 16  256x256x16 GEMM tiles (16 waves) for interior products that fill the chip in whole rounds: 1 (default) / 0
 *   key 17  ScatterConnection forward: 1 (default) = LDS-staged streaming kernel where it applies / 0 = cells-per-thread kernel
 *   key 18  channels per workgroup of that kernel (a multiple of 4 up to 64; 0, default = by LDS budget)
 *   key 21  loss finalisation folded into the last workgroup of the launch (grids up to 512): 1 (default) / 0 = finalize launch
 *   key 22  group split of the padding ops: 0 (default) = runs of equal keys / 1 = the element-level DP
 *   key 24  samples per wave of the large-batch C51 / QR-DQN forwards: 0 (default) by batch size, 1 off, 8 / 16 / 32 / 64
 *   key 25  LDS-DMA staged GEMM tiles: 1 (default) all forms / 2 the 256x256 NT tile only / 0 register staging
 *   key 26  large-batch LSTM (B >= 4096, B % 256 == 0, 768 <= H <= 1024, H % 64 == 0) row-block kernels, a bit mask: bit 0 the
 *           persistent forward, bit 3 the persistent backward (H % 128 == 0), bit 7 fence-free forward exchanges; default 9;
 *           0 = one product + one cell launch per step on the same layout (forward / backward paths mix freely)
 *   key 27  start skew of those kernels' row blocks in microseconds (0 ... 200, default 10)
 *   key 28  packed Pad1D (32 <= max_len <= 16384): 1 (default) = wave tiles in output space / 0 = the workgroup kernel
 *   key 29  mid-batch persistent LSTM forward (5 <= B <= 256, 64 <= H <= 1024, H % 16 == 0): 0 off / 1 one stream / 2 (default)
 *           two batch streams where two copies of a Wh slice fit a CU's LDS
 *   key 31  one-hot gradients of at least this many MiB are written as fill + values (default 3072; 0 = never)
 *   key 32  PPO forward in one launch: 1 (default) / 0 = three launches
 *   key 33  mid-batch persistent LSTM backward: 0 off / 1 (default) where it pays (B <= 32) / 2 every mid-batch shape
 *   key 35  16-byte quads per workgroup of the one-launch one-hot kernel (256 ... 8192; 0, default = by size)
 *   key 37  scatter owner table / chain links built inside the forward kernel: 1 (default) / 0 = index launch
 *   key 38  scatter backward in XCD-major workgroup order: 1 (default) where its pieces are below a 64-byte sector pair / 2 always / 0
 *   key 40  scatter backward by spatial tiles (a workgroup stages map rows of ALL channels and writes whole entity rows): 0 (default) by
 *           rule (16-byte pieces in the plane kernel and M * 16 >= H * W) / 1 never / 2 wherever it applies; identical results
 * hpc_rll_tune_count / hpc_rll_tune_doc enumerate the live keys (index 0 ... count-1 -> key number and a one-line description). */
int hpc_rll_tune_set(int key, int value);
int hpc_rll_tune_count(void);
const char* hpc_rll_tune_doc(int index, int* key);

/* TD(lambda) -- replaces TdLambdaForward/Backward (rl_utils/entry.h:68-77, src/rl_utils/td_lambda.cu:8-52).
 * value (T+1,B), reward (T,B), weight: mode 0 none, 1 (B,), 2 (T,B).  loss (1,) =
 * 0.5*scale*sum w (ret-V)^2; grad_buf (T,B) = d loss / d value[:T]; partials >= partials_floats(B).
 * backward: grad_value (T+1,B) = grad_loss[0] * grad_buf, last row 0. */
int hpc_rll_td_lambda_forward(const float* value, const float* reward, const float* weight, int weight_mode,
                              float* loss, float* grad_buf, float* partials, int T, int B, float gamma,
                              float lambda, float scale, void* stream);
int hpc_rll_td_lambda_backward(const float* grad_loss, const float* grad_buf, float* grad_value, int T, int B,
                               void* stream);

/* V-trace -- replaces VTraceForward/Backward (rl_utils/entry.h:131-146, src/rl_utils/vtrace.cu:8-130).
 * target/behaviour_output (T,B,N), action (T,B) int64, value (T+1,B), reward (T,B), weight (T,B) or
 * NULL.  losses (3,) = policy, value, entropy.  ws: hpc_rll_vtrace_workspace_floats(T,B) floats; its
 * first 3*T*B floats (pg / entropy coefficients, unit value gradient) must survive until backward.
 * backward recomputes the softmax from target_output (no saved (T,B,N) buffers). */
int64_t hpc_rll_vtrace_workspace_floats(int T, int B);
int hpc_rll_vtrace_forward(const float* target_output, const float* behaviour_output, const int64_t* action,
                           const float* value, const float* reward, const float* weight, float* losses, float* ws,
                           int T, int B, int N, float gamma, float lambda, float rho_clip, float c_clip,
                           float rho_pg_clip, float scale, void* stream);
int hpc_rll_vtrace_backward(const float* g_pg, const float* g_value, const float* g_ent, const float* target_output,
                            const int64_t* action, const float* ws, float* grad_target_output, float* grad_value,
                            int T, int B, int N, void* stream);

/* UPGO -- replaces UpgoForward/Backward (rl_utils/entry.h:148-156, src/rl_utils/upgo.cu:8-70).
 * target_output (T,B,N), rho (T,B), action (T,B), reward (T,B), value (T+1,B); loss (1,). */
int64_t hpc_rll_upgo_workspace_floats(int T, int B);
int hpc_rll_upgo_forward(const float* target_output, const float* rho, const int64_t* action, const float* reward,
                         const float* value, float* loss, float* ws, int T, int B, int N, float scale, void* stream);
int hpc_rll_upgo_backward(const float* g, const float* target_output, const int64_t* action, const float* ws,
                          float* grad_target_output, int T, int B, int N, void* stream);

/* PPO -- replaces PPOForward/Backward (rl_utils/entry.h:158-165, src/rl_utils/ppo.cu:8-111).
 * logits (B,N), action (B,), value_new/old, adv, ret, weight (B,) (weight NULL = ones).
 * out5 = policy_loss, value_loss, entropy_loss, approx_kl, clipfrac.  dual_clip < 1 disables dual clip
 * (the reference encodes None as 0: hpc_rll/rl_utils/ppo.py:136-137). */
int64_t hpc_rll_ppo_workspace_floats(int B);
int hpc_rll_ppo_forward(const float* logits_new, const float* logits_old, const int64_t* action,
                        const float* value_new, const float* value_old, const float* adv, const float* ret,
                        const float* weight, float* out5, float* ws, int B, int N, float clip_ratio,
                        int use_value_clip, float dual_clip, float scale, void* stream);
int hpc_rll_ppo_backward(const float* g_policy, const float* g_value, const float* g_ent, const float* logits_new,
                         const int64_t* action, const float* ws, float* grad_logits_new, float* grad_value_new,
                         int B, int N, void* stream);

/* q n-step TD (rescale=0) / with value rescaling (rescale=1) -- replaces QNStepTd{,Rescale}Forward/Backward
 * (rl_utils/entry.h:89-109).  q,next_n_q (B,N); action,next_n_action (B,) int64; reward (nstep,B); done,
 * weight (B,) float (weight NULL = ones).  loss (1,), td_err (B,), grad_buf (B,). */
int hpc_rll_q_nstep_td_forward(const float* q, const float* next_n_q, const int64_t* action,
                               const int64_t* next_n_action, const float* reward, const float* done,
                               const float* weight, float* loss, float* td_err, float* grad_buf, float* partials,
                               int nstep, int B, int N, float gamma, int rescale, float scale, void* stream);
int hpc_rll_q_nstep_td_backward(const float* grad_loss, const float* grad_buf, const int64_t* action, float* grad_q,
                                int B, int N, void* stream);

/* dist (C51) n-step TD -- replaces DistNStepTdForward/Backward (rl_utils/entry.h:79-87).
 * dist,next_n_dist (B,N,n_atom); buf (B,n_atom) = unit gradient wrt dist[b,a_b,:]. */
int hpc_rll_dist_nstep_td_forward(const float* dist, const float* next_n_dist, const int64_t* action,
                                  const int64_t* next_n_action, const float* reward, const float* done,
                                  const float* weight, float* loss, float* td_err, float* buf, float* partials,
                                  int nstep, int B, int N, int n_atom, float gamma, float v_min, float v_max,
                                  float scale, void* stream);
int hpc_rll_dist_nstep_td_backward(const float* grad_loss, const float* buf, const int64_t* action,
                                   float* grad_dist, int B, int N, int n_atom, void* stream);

/* IQN n-step TD -- replaces IQNNStepTDErrorForward/Backward (rl_utils/entry.h:111-119).
 * q (tau,B,N), next_n_q (tau',B,N), replay_quantiles (tau,B), value_gamma (B,) or NULL (= gamma^nstep);
 * buf (B,tau) = unit gradient wrt q[:,b,a_b]. */
int hpc_rll_iqn_nstep_td_forward(const float* q, const float* next_n_q, const int64_t* action,
                                 const int64_t* next_n_action, const float* reward, const float* done,
                                 const float* replay_quantiles, const float* weight, const float* value_gamma,
                                 float* loss, float* td_err, float* buf, float* partials, int tau, int tau_prime,
                                 int nstep, int B, int N, float gamma, float kappa, float scale, void* stream);
int hpc_rll_iqn_nstep_td_backward(const float* grad_loss, const float* buf, const int64_t* action, float* grad_q,
                                  int tau, int B, int N, void* stream);
/* ABI 6 (not in the reference): the same loss with the quantile axis INNERMOST -- q (B,N,tau), next_n_q (B,N,tau'), grad_q
 * (B,N,tau), the layout of the QR-DQN op (rl_utils/entry.h:121-129); replay_quantiles stays (tau,B), buf (B,tau).  A sample's
 * quantiles are one contiguous row instead of tau values a (B,N) plane apart: 2 cache lines per sample instead of 2 tau
 * (iqn_nstep_td_error_kernel.h:11-70 reads the (tau,B,N) layout). */
int hpc_rll_iqn_nstep_td_forward_bnt(const float* q, const float* next_n_q, const int64_t* action,
                                     const int64_t* next_n_action, const float* reward, const float* done,
                                     const float* replay_quantiles, const float* weight, const float* value_gamma,
                                     float* loss, float* td_err, float* buf, float* partials, int tau, int tau_prime,
                                     int nstep, int B, int N, float gamma, float kappa, float scale, void* stream);
int hpc_rll_iqn_nstep_td_backward_bnt(const float* grad_loss, const float* buf, const int64_t* action, float* grad_q,
                                      int tau, int B, int N, void* stream);

/* QR-DQN n-step TD -- replaces QRDQNNStepTDErrorForward/Backward (rl_utils/entry.h:121-129).
 * q,next_n_q (B,N,tau); tau_value = the `tau` the caller passes to the oracle (the reference kernel hard
 * codes the integer count, qrdqn_nstep_td_error_kernel.h:60); buf (B,tau). */
int hpc_rll_qrdqn_nstep_td_forward(const float* q, const float* next_n_q, const int64_t* action,
                                   const int64_t* next_n_action, const float* reward, const float* done,
                                   const float* weight, const float* value_gamma, float* loss, float* td_err,
                                   float* buf, float* partials, int tau, int nstep, int B, int N, float gamma,
                                   float tau_value, float scale, void* stream);
int hpc_rll_qrdqn_nstep_td_backward(const float* grad_loss, const float* buf, const int64_t* action, float* grad_q,
                                    int tau, int B, int N, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pad / Unpad of a ragged list of n contiguous fp32 tensors of rank 1..3 -- replaces Pad{1,2,3}DForward,
 * GroupPad{1,2,3}DForward, Unpad{1,2,3}DForward (rl_utils/entry.h:10-59, src/rl_utils/padding.cu:111-582).
 * A rank-1 tensor of length L is described as (d0,d1,d2) = (1,1,L), rank-2 (a,b) as (1,a,b).
 *   pad  : table (device, n x 4 int64) = {source pointer, d0, d1, d2}; new_x (n,m0,m1,m2) fp32 gets the
 *          data and `value` elsewhere; mask (same shape, int32) gets 1 inside and `value` outside.
 *   unpad: table (device, n x 4 int64) = {offset of tensor i in `flat` (elements), d0, d1, d2}, offsets
 *          ascending; flat (total floats) receives the tensors back to back.
 * Group padding = one pad call per group (the grouping policy is host logic, below).
 * ------------------------------------------------------------------------------------------ */
int hpc_rll_pad_forward(const int64_t* table, float* new_x, int32_t* mask, int64_t n, int m0, int m1, int m2,
                        int value, void* stream);
int hpc_rll_unpad_forward(const float* padded, const int64_t* table, float* flat, int64_t n, int64_t total,
                          int m0, int m1, int m2, void* stream);
/* Packed (CSR-style) ragged input: lengths (n,) int64 on the device -> the (n,4) table the two kernels above take,
 * row i = {base + stride * sum(lengths[:i]), 1, 1, lengths[i]} (pad: base = address of the flat buffer, stride = 4;
 * unpad: base = 0, stride = 1).  A device-side exclusive scan: no host loop, no synchronisation.
 * scratch: hpc_rll_packed_table_scratch_int64(n) int64. */
int64_t hpc_rll_packed_table_scratch_int64(int64_t n);
int hpc_rll_packed_table(const int64_t* lengths, int64_t n, int64_t base, int64_t stride, int64_t* table,
                         int64_t* scratch, void* stream);
/* Pad1DForward / Unpad1DForward (src/rl_utils/padding.cu:111-140, 228-260; kernels padding_kernel.h:92-127) for PACKED
 * rows: the two directions with the rows' contiguity exploited (a workgroup's rows are ONE span of the flat buffer,
 * staged through LDS with 16-byte accesses).  `table` as built by hpc_rll_packed_table: pad -- base = address of `flat`,
 * stride 4; unpad -- base 0, stride 1.  Same results as hpc_rll_pad_forward / hpc_rll_unpad_forward with m0 = m1 = 1
 * (to which they fall back when a pointer is not aligned for 16-byte stores). */
int hpc_rll_pad1d_packed_forward(const float* flat, const int64_t* table, float* new_x, int32_t* mask, int64_t n,
                                 int max_len, int value, void* stream);
int hpc_rll_unpad1d_packed_forward(const float* padded, const int64_t* table, float* flat, int64_t n, int64_t total,
                                   int max_len, void* stream);
/* Group-split policies over a list sorted by numel (host code; padding.cu:8-108).  sizes: n x dim int32.
 * Write <= `group` rows of `dim` ints to group_shapes and <= group+1 boundaries to positions; return the
 * number of groups (>= 1) or a negative HPC_RLL_E* code.  oracle = the O(group * n^2) DP minimising padded
 * elements (ties -> smallest split point); sample = random cuts (deterministic for a given seed). */
int hpc_rll_oracle_split_group(const int32_t* sizes, int n, int dim, int group, int32_t* group_shapes,
                               int32_t* positions);
int hpc_rll_sample_split_group(const int32_t* sizes, int n, int dim, int group, uint64_t seed,
                               int32_t* group_shapes, int32_t* positions);
/* Grouped padding of PACKED 1-D rows entirely on the device (what hpc_rll/rl_utils/padding.py:20-45 does on the host
 * with a python sorted() + the split policies above + one pad per group; SURVEY.md 8f-3: ~1M rows at configs[4]).
 * lengths (n,) int64 on the device, every length in [0, max_len], max_len <= 16384, group <= 63.
 *   plan  : histogram of the lengths -> runs of equal lengths -> the split policy on the runs (mode 0 = oracle: the same
 *           cuts as hpc_rll_oracle_split_group incl. its tie rule; mode 1 = sample with `seed`) -> `plan`, 3*group+4
 *           int64 on the device: [0] number of groups ng, [1] status (1 = a length was outside [0,max_len] and clamped),
 *           [2 .. 2+group] cut positions in sorted order, [3+group .. 3+2*group) group widths,
 *           [3+2*group .. 4+3*group) offsets of the groups in the concatenated output (ng+1 used);
 *           and a STABLE radix sort of (length, row) -> order (n,) int64: order[p] = original row of sorted position p
 *           (ascending length, original order among equal lengths = python's sorted()).
 *           ws: hpc_rll_pad1d_group_workspace_int64(n, max_len, group) int64.  No host synchronisation.
 *           Cost of the plan step: the histogram and the sort stream `lengths` (HBM-bound); the oracle policy
 *           itself is a DP over the D <= max_len+1 DISTINCT lengths that runs in ONE workgroup, group * D^2 / 1024 steps of
 *           ~10 ns: microseconds at configs[4] (D <= 96: len ~ U[32,128)), ~2 ms at D = 4096 with group = 8, ~0.25 s at the
 *           limits D = 16385, group = 63 -- there the host's divide-and-conquer DP (hpc_rll_oracle_split_group on the
 *           histogram's runs) is the faster route; the prefix scan of the sort is single-workgroup too: n <= 2^28.
 *   forward: all groups in ONE launch.  table = hpc_rll_packed_table rows of the ORIGINAL order; out / mask hold the
 *           groups back to back (group g: (cuts[g+1]-cuts[g]) rows of width[g] at offset[g]); total_out = offset[ng]. */
int64_t hpc_rll_pad1d_group_workspace_int64(int64_t n, int max_len, int group);
int hpc_rll_pad1d_group_plan(const int64_t* lengths, int64_t n, int max_len, int group, int mode, uint64_t seed,
                             int64_t* ws, int64_t* plan, int64_t* order, void* stream);
int hpc_rll_pad1d_group_forward(const int64_t* table, const int64_t* order, const int64_t* plan, int group, float* out,
                                int32_t* mask, int64_t total_out, int value, void* stream);

/* ScatterConnection -- replaces ScatterConnectionForward/Backward (torch_utils/network/entry.h:21-29,
 * src/torch_utils/network/scatter_connection.cu:8-73).  x (B,M,N) fp32, location (B,M,2) int64 (y,x),
 * out (B,N,H,W) fp32 (fully written: no pre-zeroing needed); add=0: "cover" (the largest m at a cell wins,
 * like the CPU oracle), add=1: sum in ascending m.  ws: hpc_rll_scatter_workspace_ints(B,M,H,W) int32 of scratch
 * (contents unspecified afterwards; untouched where the output kernel builds its tables in LDS, tune key 37).
 * backward: grad_x[b,m,:] = grad_out[b,:,y,x] for every entity (also the covered ones). */
int64_t hpc_rll_scatter_workspace_ints(int B, int M, int H, int W);
int hpc_rll_scatter_connection_forward(const float* x, const int64_t* location, float* out, int32_t* ws, int B,
                                       int M, int N, int H, int W, int add, void* stream);
int hpc_rll_scatter_connection_backward(const float* grad_out, const int64_t* location, float* grad_x, int B, int M,
                                        int N, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm-LSTM -- replaces LstmForward/LstmBackward (torch_utils/network/entry.h:11-19,
 * src/torch_utils/network/lstm.cu:29-379).  x (S,B,I); h0,c0 (L,B,H); wx = concat_l row-major [in_l,4H]
 * (in_0 = I, in_l = H); wh = L x [H,4H]; bias (L,4H); ln_gamma/ln_beta (L, 2*4H) = [x-half | h-half];
 * gate order i,f,o,u.  Outputs y (S,B,H), hn,cn (L,B,H).  ws: hpc_rll_lstm_workspace_floats(...) floats,
 * must be kept (unmodified) from forward to backward.  All GEMMs are exact fp32 on the matrix cores.
 * dropout_p in [0,1): inter-layer dropout, mask = stateless hash of (seed, layer, element).
 * backward: dy (S,B,H), dhn, dcn (L,B,H) may each be NULL (= zero).  Unlike the reference, gradients
 * flowing in through hn / cn are honoured.  dx may be NULL when the gradient of x is not needed: the
 * S*B x I x 4H input-gradient product of layer 0 is then skipped.
 * ------------------------------------------------------------------------------------------ */
int64_t hpc_rll_lstm_workspace_floats(int S, int B, int I, int H, int L, float dropout_p);
/* Float offset of the last layer's h sequence inside ws.  It IS y (the reference keeps it the same way: ym[L-1],
 * rnn.py:27-31): a forward called with y = ws + offset writes y in place and skips the (S,B,H) copy; any other y is
 * filled by a copy as before.  -1 on invalid arguments. */
int64_t hpc_rll_lstm_workspace_y_offset(int S, int B, int I, int H, int L, float dropout_p);
int hpc_rll_lstm_forward(const float* x, const float* h0, const float* c0, const float* wx, const float* wh,
                         const float* bias, const float* ln_gamma, const float* ln_beta, float* y, float* hn,
                         float* cn, float* ws, int S, int B, int I, int H, int L, float dropout_p, uint64_t seed,
                         void* stream);
int hpc_rll_lstm_backward(const float* dy, const float* dhn, const float* dcn, const float* x, const float* h0,
                          const float* c0, const float* wx, const float* wh, const float* ln_gamma, float* ws,
                          float* dx, float* dh0, float* dc0, float* dwx, float* dwh, float* dbias, float* dln_gamma,
                          float* dln_beta, int S, int B, int I, int H, int L, float dropout_p, uint64_t seed,
                          void* stream);
/* ABI 4: the pair a framework binding uses.  y is the caller's own (S,B,H) tensor AND the last layer's saved h sequence
 * (the reference keeps the same array in both roles: ym[L-1], rnn.py:27-31, read again by LstmBackward lstm.cu:352-370):
 * the forward's cells write y directly -- no (S,B,H) copy and no view of the workspace handed out, so a holder of y pins
 * S*B*H floats, not the workspace -- and the SAME y is passed to the backward, which reads the sequence from it.  y must
 * not be modified in between (the binding's job to detect; torch's saved-tensor version counter does). */
int hpc_rll_lstm_forward_y(const float* x, const float* h0, const float* c0, const float* wx, const float* wh,
                           const float* bias, const float* ln_gamma, const float* ln_beta, float* y, float* hn,
                           float* cn, float* ws, int S, int B, int I, int H, int L, float dropout_p, uint64_t seed,
                           void* stream);
int hpc_rll_lstm_backward_y(const float* dy, const float* dhn, const float* dcn, const float* x, const float* h0,
                            const float* c0, const float* wx, const float* wh, const float* ln_gamma, const float* y,
                            float* ws, float* dx, float* dh0, float* dc0, float* dwx, float* dwh, float* dbias,
                            float* dln_gamma, float* dln_beta, int S, int B, int I, int H, int L, float dropout_p,
                            uint64_t seed, void* stream);
/* Diagnostic: the kernels the most recent hpc_rll_lstm_forward* call of this process ran its recurrence on.  0 = one product
 * + one cell launch per step (what src/torch_utils/network/lstm.cu:145-161 does with three launches), 1 = per-layer
 * persistent kernels (B <= 4), 2 = layer wavefront (B <= 4, L >= 2), 3 = step kernels on gate-interleaved pre-activations
 * (large batch), 4 = persistent row-block kernel (large batch, tune key 26), 5 = persistent mid-batch kernel (5 <= B <= 256,
 * tune key 29); -1 = no forward yet. */
int hpc_rll_lstm_last_forward_path(void);
int hpc_rll_lstm_last_backward_path(void);   /* the same for the most recent hpc_rll_lstm_backward* call (its last layer; 5 = mid-batch kernel, key 33) */
/* Asynchronous status of the persistent small-batch LSTM kernels (B <= 4).  Their workgroups exchange data through
 * memory and must all be resident at once; that is checked against the runtime's occupancy figure at dispatch and
 * launches of one process are serialised per device, but ANOTHER PROCESS holding compute units for seconds can still
 * starve one.  The kernel then gives up (no trap, no hang), and -- like an asynchronous HIP error -- every later
 * hpc_rll_lstm_* call returns HPC_RLL_ETIMEOUT (the results of the launch that timed out are invalid) until
 * hpc_rll_clear_async_error(), after which the process uses the step kernels only.  hpc_rll_async_error() reads the
 * status without touching the device. */
int hpc_rll_async_error(void);
int hpc_rll_clear_async_error(void);
/* Test hooks, not operators.  occupy_device: launch `blocks` workgroups of 1024 threads and 80 KB of LDS (two of them
 * fill a CU completely; 0 = 2 per CU = the whole device) that spin for ~ms milliseconds on `stream`.  set_persist_spin_limit: polls a persistent LSTM kernel of the current
 * device waits before it gives up (0 = the shipped value, ~seconds). */
int hpc_rll_test_occupy_device(int ms, int blocks, void* stream);
int hpc_rll_test_set_persist_spin_limit(int64_t polls);
/* Exact-fp32 MFMA GEMM used by the LSTM, exposed for tests/benchmarks: C (M,N; row stride ldc) (+)= A * B with
 * A(m,k) = A[m*a_sm + k*a_sk], B(k,n) = B[k*b_sk + n*b_sn]. */
int hpc_rll_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int64_t a_sm, int64_t a_sk,
                     int64_t b_sk, int64_t b_sn, int64_t ldc, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * hpc_models: AlphaStar actor-critic inference helpers, forward only -- replaces actor_critic_update_ae,
 * actor_critic_lstm_activation, actor_critic_pre_sample (include/hpc/rll/cuda/models/entry.h:11-21,
 * src/models/actor_critic.cu:8-83).
 *   update_ae      : ae (B,D) += key_embeddings[b, sample_entity[b], :] unless sample_entity[b] == entity_num[b]
 *   lstm_activation: gates = ih + hh + bias (B,4H; order i,f,g,o); c (B,H) updated in place, h (B,H) written
 *   pre_sample     : out (B,E) = mask ? dot(mat[b,e,:], vec[b,:]) / div : mask_value / div   (mask: 1 byte per entry)
 * ------------------------------------------------------------------------------------------ */
int hpc_rll_actor_critic_update_ae(const float* key_embeddings, const int64_t* sample_entity,
                                   const int64_t* entity_num, float* autoregressive_embedding, int64_t B, int64_t E,
                                   int64_t D, void* stream);
int hpc_rll_actor_critic_lstm_activation(const float* ih, const float* hh, const float* bias, float* h, float* c,
                                         int64_t B, int64_t H, void* stream);
int hpc_rll_actor_critic_pre_sample(const float* mat, const float* vec, const uint8_t* mask, float* out, int64_t B,
                                    int64_t E, int64_t H, float mask_value, float div_factor, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HPC_RLL_HIP_H_ */
