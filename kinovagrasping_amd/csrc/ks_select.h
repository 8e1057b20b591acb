// ks_select.h -- the per-env action-selection rule shared by k_select_action (ks_rollout.hip) and the fused
// actor + selection kernel (ks_mlp.hip): check_grasp latch (expert_data.py:559-593), exploration noise + clip
// (main_DDPGfD.py:443-446), scripted lift action (main_DDPGfD.py:275-290, 945-947).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace krsel {

constexpr int S = 82, A = 4;

// scripted lift: wrist_lift_velocity, finger_lift_velocity x3 (main_DDPGfD.py:945-947)
__device__ __forceinline__ float lift_action(int k) { return k == 0 ? 0.6f : 0.5f; }

// env i: pi = actor output (4), nz = N(0,1) noise (4).  Updates ready (latched), writes action / action_t / lifting.
__device__ __forceinline__ void select_one(int i, int n, const float* pi, const float* nz, const float* __restrict__ obs,
                                           const float* __restrict__ prev_obs, const uint8_t* __restrict__ has_prev, const int64_t* __restrict__ t,
                                           uint8_t* ready, float sigma, float max_action, int skip_steps, float* __restrict__ action,
                                           float* __restrict__ action_t, uint8_t* __restrict__ lifting) {
    // no fma contraction here: the products are rounded before the sums, bit-identical to the torch expressions
#pragma clang fp contract(off)
    // check_grasp on obs[9:17]: x of the three distal fingertips (columns 9, 12, 15), per substep (frame_skip 15)
    const float* o = obs + (long)i * S;
    const float* p = prev_obs + (long)i * S;
    float d = fabsf(p[9] - o[9]) / 15.0f;
    d += fabsf(p[12] - o[12]) / 15.0f;
    d += fabsf(p[15] - o[15]) / 15.0f;
    const bool chk = d < 0.0002f && (t[i] + 1 >= skip_steps) && has_prev[i] != 0;
    const bool rdy = ready[i] != 0 || chk;
    ready[i] = rdy;
    lifting[i] = rdy;
#pragma unroll
    for (int k = 0; k < A; k++) {
        float a = pi[k] + nz[k] * sigma;
        a = fminf(fmaxf(a, 0.0f), max_action);
        a = rdy ? lift_action(k) : a;
        action[(long)i * A + k] = a;
        action_t[(long)k * n + i] = a;
    }
}

// Philox4x32-10 (Salmon et al. 2011): counter (c0..c3), key (k0, k1) -> 4 x 32 random bits
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* r) {
#pragma unroll
    for (int round = 0; round < 10; round++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

// 4 independent N(0,1) draws for (seed, step, env): Philox bits -> uniforms in (0,1) -> Box-Muller
__device__ __forceinline__ void normal4(uint64_t seed, uint64_t step, uint32_t env, float* z) {
    uint32_t r[4];
    philox4x32(env, (uint32_t)step, (uint32_t)(step >> 32), 0x4b52u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float u0 = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = ((float)(r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(r[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = ((float)(r[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float ra = sqrtf(-2.0f * __logf(u0)), rb = sqrtf(-2.0f * __logf(u2));
    float sa, ca, sb, cb;
    __sincosf(6.283185307179586f * u1, &sa, &ca);
    __sincosf(6.283185307179586f * u3, &sb, &cb);
    z[0] = ra * ca; z[1] = ra * sa; z[2] = rb * cb; z[3] = rb * sb;
}

}  // namespace krsel
