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

/* ABI version, bumped on any signature change. */
int hpc_rll_abi_version(void);
/* Human readable message for a status returned by any entry point (static storage). */
const char* hpc_rll_status_string(int status);

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

#ifdef __cplusplus
}
#endif
#endif /* HPC_RLL_HIP_H_ */
