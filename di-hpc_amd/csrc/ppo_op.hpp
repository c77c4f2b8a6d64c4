// ppo_op.hpp -- the per-sample arithmetic of the PPO loss (hpc_rll/origin/ppo.py:51-80; replaces ppo_kernel.h:12-283 under
// /root/reference), shared by the three-launch forward (sample_ops.hip: two categorical launches + sample_kernel<PpoOp>) and
// the fused one-launch forward (categorical.hip: ppo_fwd_fused_kernel).
#pragma once
#include <hip/hip_runtime.h>

namespace hpc_rll {

struct PpoOp {
    static constexpr int NACC = 5;
    const float *logp_new, *ent, *logp_old, *value_new, *value_old, *adv, *ret, *weight;
    float *coef_logp, *coef_ent, *gv_unit;
    float clip, dual_clip, scale;
    int use_value_clip;
    struct In { float w, a, vn, r, vo; };   // the per-sample inputs, loadable before the heads' statistics are known
    __device__ __forceinline__ In load(long i) const {
        In v;
        v.w = weight ? weight[i] : 1.f;
        v.a = adv[i];
        v.vn = value_new[i];
        v.r = ret[i];
        v.vo = use_value_clip ? value_old[i] : 0.f;
        return v;
    }
    __device__ __forceinline__ void operator()(long i, float (&acc)[NACC]) const { apply(i, load(i), logp_new[i], ent[i], logp_old[i], acc); }
    __device__ __forceinline__ void apply(long i, const In& in, float lpn, float entv, float lpo, float (&acc)[NACC]) const {
        const float w = in.w;
        const float a = in.a;
        const float ratio = expf(lpn - lpo);
        const float s1 = ratio * a;
        const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
        const float s2 = rc * a;
        // min(s1, s2): s1 wins ties (identical value and identical derivative whenever they tie inside the clip range)
        float inner = s1, dinner = s1;  // d inner / d logp_new = ratio * adv on the s1 branch
        if (s2 < s1) { inner = s2; dinner = (rc == ratio) ? s1 : 0.f; }
        if (dual_clip >= 1.f) {          // reference encodes "None" as 0 (rl_utils/ppo.py:136-137, ppo_kernel.h:188)
            const float d = dual_clip * a;
            if (d > inner) { inner = d; dinner = 0.f; }
        }
        acc[0] -= inner * w;
        coef_logp[i] = -dinner * w * scale;
        const float vn = in.vn, r = in.r;
        float v = (r - vn) * (r - vn);
        float dv = -(r - vn);            // d(0.5 v)/d value_new on the unclipped branch
        if (use_value_clip) {
            const float vo = in.vo;
            const float dvo = vn - vo;
            const bool saturated = dvo > clip || dvo < -clip;   // d vclip / d value_new = 0 only when the clamp is active
            const float vc = vo + fminf(fmaxf(dvo, -clip), clip);
            const float v2 = (r - vc) * (r - vc);
            // NB: with an inactive clamp vo + (vn - vo) can differ from vn by an ulp in fp32, so v2 may exceed v
            // although mathematically equal; the gradient must then still flow (found by tests/test_fuzz_gpu.py)
            if (v2 > v) { v = v2; dv = saturated ? 0.f : -(r - vc); }
        }
        acc[1] = fmaf(v, w, acc[1]);
        gv_unit[i] = dv * w * scale;
        acc[2] = fmaf(entv, w, acc[2]);
        coef_ent[i] = w * scale;
        acc[3] += lpo - lpn;
        acc[4] += (ratio > 1.f + clip || ratio < 1.f - clip) ? 1.f : 0.f;
    }
};

}  // namespace hpc_rll
