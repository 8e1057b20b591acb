// ks_mlp.hip -- fused 3-layer MLP forward on the matrix cores (C ABI: include/kinova_rollout.h, kr_mlp3_forward).
//
// The reference's actor and critic (DDPGfD.py:17-50) are  out = f(W3 relu(W2 relu(W1 x + b1) + b2) + b3)  with
// 82 / 86 inputs, two hidden layers (256-256 in BASELINE, 400-300 in the reference) and 4 / 1 outputs.  As three
// library GEMMs + bias/activation kernels that is 8-11 launches of a few microseconds each, on the critical path of
// every env-step (the action of step t+1 needs the observation of step t).  Here it is ONE launch:
//
//   * a workgroup (4 waves, one per SIMD) owns 16 batch rows and computes all three layers for them; the activations
//     never leave the CU (LDS), the weights (<= 0.7 MB, L2 resident) stream through the MFMA A operand;
//   * exact fp32 on v_mfma_f32_16x16x4_f32, in the TRANSPOSED orientation  H^T = W X^T : A = W (16 output features x
//     4 k), B = X^T (4 k x 16 batch rows), D = 16 features x 16 rows.  A lane of D holds features 4q..4q+3 (q = lane>>4)
//     of batch row n = lane&15 - exactly the 4 k-values the same lane must supply as the B operand of the next
//     layer's four MFMAs of a 16-wide k-step.  So a layer's output quad is stored as one float4 at [tile*4 + q][n] and
//     read back from the same slot: no transposes, no bank conflicts (16 consecutive float4 per quarter wave);
//   * the waves split the output tiles of layers 1 and 2 and the k-steps of layer 3 (partial sums through LDS).
//
// Arithmetic: every output is a k-ordered fp32 fma chain (MFMA f32 is bitwise an fmaf chain), so results differ
// from the library GEMMs only by summation order (~1e-7 relative); tests/test_gpu_parity.py checks against torch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/kinova_rollout.h"
#include "../../include/kinova_sim.h"
#include "ks_mlp_tile.h"
#include "ks_select.h"

namespace {

using namespace kmlp;

// occupancy target of the 2- / 4-wave split learner kernels: 3 waves per SIMD = up to 168 registers (156 used, no spills); the
// experiment switch 4 caps them at 128 (140 B of scratch per lane) so that they fit beside k_rollout's 368
#ifndef KS_SPLIT_WAVES_PER_EU
#define KS_SPLIT_WAVES_PER_EU 3
#endif

// Epilogue of the fused actor + action-selection launch (kr_actor_select): everything k_select_action takes, plus the
// noise source (a tensor of N(0,1) draws, or the in-kernel counter-based generator keyed by (seed, rng_state[0], env))
struct SelectArgs {
    const float* obs; const float* prev_obs; const uint8_t* has_prev; const int64_t* t; uint8_t* ready;
    const float* noise; unsigned long long seed; int64_t* rng_state;
    float sigma, max_action; int skip_steps;
    float* action; float* action_t; uint8_t* lifting;
};

// NT1 / NT2: 16-feature tiles of the two hidden layers.  x is [n][ldx] with the first in_a columns from xa and, when
// xb != nullptr, the next in_b columns from xb ([n][ldb]) - the critic's cat([state, action]) without materialising it.
template <int NT1, int NT2, bool VEC, bool SEL>
__global__ __launch_bounds__(64 * NW) void k_mlp3(int n, int in_a, int in_b, int h1, int h2, int out_dim, const float* __restrict__ xa, int lda,
                                              const float* __restrict__ xb, int ldb, const float* __restrict__ W1, const float* __restrict__ b1,
                                              const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ W3,
                                              const float* __restrict__ b3, int act, float scale, float* __restrict__ out, float* __restrict__ h1_out,
                                              float* __restrict__ h2_out, SelectArgs sel) {
    __shared__ f32x4 H1[NT1 * 4][ROWS];
    __shared__ f32x4 H2[NT2 * 4][ROWS];
    __shared__ f32x4 P[NW][ROWS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nn = lane & 15;
    const int row = blockIdx.x * ROWS + nn;
    const bool row_ok = row < n;
    unsigned long long rng_step = 0;
    if (SEL && sel.rng_state) rng_step = (unsigned long long)sel.rng_state[0];      // read by every workgroup before any of them finishes
    f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    if (mlp3_rows16<NT1, NT2, VEC>(wave, lane, row_ok ? (long)row : -1L, in_a, in_b, h1, h2, out_dim, xa, lda, xb, ldb, W1, b1, W2, b2, W3, h1_out, h2_out,
                                   H1, H2, P, z4)) {
        const float z[4] = {z4.x, z4.y, z4.z, z4.w};
        float y[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < out_dim; i++) {
            y[i] = z[i] + b3[i];
            if (act == KR_ACT_SIGMOID) y[i] = scale / (1.f + __expf(-y[i]));
            if (out) out[(long)row * out_dim + i] = y[i];
        }
        if (SEL) {
            float nz[4];
            if (sel.noise) {
#pragma unroll
                for (int k = 0; k < 4; k++) nz[k] = sel.noise[(long)row * 4 + k];
            } else {
                krsel::normal4(sel.seed, rng_step, (uint32_t)row, nz);
            }
            krsel::select_one(row, n, y, nz, sel.obs, sel.prev_obs, sel.has_prev, sel.t, sel.ready, sel.sigma, sel.max_action, sel.skip_steps,
                              sel.action, sel.action_t, sel.lifting);
        }
    }
    if (SEL && sel.rng_state) {
        // the LAST workgroup to finish advances the step counter: every workgroup has read it by then
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned ticket = atomicAdd((unsigned*)(sel.rng_state + 1), 1u);
            if (ticket == gridDim.x - 1) {
                sel.rng_state[1] = 0;
                sel.rng_state[0] = (int64_t)(rng_step + 1);
            }
        }
    }
}

// ---- the same network WITHOUT LDS: one wave = 16 batch rows, all layers in registers.  Layer 1's output quads (NT1
// float4 registers) are the B operands of layer 2; every layer-2 output quad feeds layer 3's MFMAs right away, so h2
// is never stored.  The kernel is capped at 168 registers per lane (3 waves per SIMD): k_env_step holds all of a
// CU's LDS and 344 of the 512 registers of every SIMD lane, so these waves can be resident BESIDE it and use the
// matrix pipes and issue slots the stepping kernel leaves idle - the learner's forward-only passes run in its shadow
// instead of waiting for its workgroups to retire.  Slower per wave than the LDS kernel (no split of the tiles over
// four waves), which does not matter there.
template <int NT1, int NT2, bool VEC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_mlp3_wave(
    int n, int in_a, int in_b, int h1, int h2, int out_dim, const float* __restrict__ xa, int lda, const float* __restrict__ xb, int ldb,
    const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ W3, const float* __restrict__ b3, int act, float scale, float* __restrict__ out, float* __restrict__ h1_out,
    float* __restrict__ h2_out) {
    const int lane = threadIdx.x & 63, nn = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * ROWS + nn;
    const bool row_ok = row < n;
    const int in_dim = in_a + in_b;
    // All operand reads are raw buffer loads: ONE 32-bit lane offset per matrix, the tile / k-step advance in the
    // wave-uniform scalar offset, out-of-range reads return 0 in hardware.  The flat-load version spent 5 VALU + 7 SALU
    // instructions (address arithmetic, bounds branches) per MFMA - issue slots this kernel shares with the stepping
    // kernel's wave on the same SIMD.  (The host checks h1 % 16 == h2 % 16 == 0 for this variant.)
    const auto rW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W1), 0, h1 * in_dim * 4, 0x00020000);
    const auto rW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W2), 0, h2 * h1 * 4, 0x00020000);
    const auto rW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W3), 0, out_dim * h2 * 4, 0x00020000);
    const auto rXa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xa), 0, ((n - 1) * lda + in_a) * 4, 0x00020000);
    const auto rXb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb ? xb : xa), 0, xb ? ((n - 1) * ldb + in_b) * 4 : 0, 0x00020000);
#define KS_LDF(rsrc, voff, soff) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0))
    f32x4 h1r[NT1];
    {
        // B operands of layer 1: the 16 input rows, k = 16 s + 4 q + j.  A row beyond n is out of the buffer's range (0).
        // (the range check covers the LANE offset only, not the scalar offset: an element that does not exist gets an
        //  out-of-range lane offset instead of relying on the sum)
        constexpr int OOR = 0x7ffffff0;
        f32x4 bx[KS_IN_MAX];
        const int oa = row * lda * 4 + 16 * q, ob = (row * ldb + 4 * q - in_a) * 4;
#pragma unroll
        for (int s = 0; s < KS_IN_MAX; s++) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = 16 * s + 4 * q + j;
                const float fa = KS_LDF(rXa, (row_ok && k < in_a) ? oa : OOR, (16 * s + j) * 4);
                const float fb = KS_LDF(rXb, (row_ok && k >= in_a && k < in_dim) ? ob + (16 * s + j) * 4 : OOR, 0);
                v[j] = k < in_a ? fa : fb;
            }
            bx[s] = f32x4{v[0], v[1], v[2], v[3]};
        }
        // A operands: W1[16 t + nn][16 s + 4 q + j], zero beyond in_dim
        const int o1 = (nn * in_dim + 4 * q) * 4;
#pragma unroll
        for (int t = 0; t < NT1; t++) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS_IN_MAX; s++) {
                const int so = (16 * t * in_dim + 16 * s) * 4, k0 = 16 * s + 4 * q;
                const f32x4 w = {KS_LDF(rW1, k0 < in_dim ? o1 : OOR, so), KS_LDF(rW1, k0 + 1 < in_dim ? o1 : OOR, so + 4),
                                 KS_LDF(rW1, k0 + 2 < in_dim ? o1 : OOR, so + 8), KS_LDF(rW1, k0 + 3 < in_dim ? o1 : OOR, so + 12)};
                if (s & 1) acc1 = mfma4(w, bx[s], acc1);
                else acc0 = mfma4(w, bx[s], acc0);
            }
            h1r[t] = bias_relu(acc0 + acc1, b1, t * 16 + 4 * q, h1);
            if (h1_out && row_ok) *(f32x4*)(h1_out + (long)row * h1 + t * 16 + 4 * q) = h1r[t];
        }
    }
    f32x4 acc3 = {0.f, 0.f, 0.f, 0.f};
    const int o2 = (nn * h1 + 4 * q) * 4;                       // W2[16 t + nn][16 s + 4 q ..]: 16-byte reads (h1 % 4 == 0)
    const int o3 = (nn * h2 + 4 * q) * 4;                       // W3[nn][16 t + 4 q ..]: rows >= out_dim are out of range (0)
#pragma unroll 1
    for (int t = 0; t < NT2; t++) {
        f32x4 w[NT1];
#pragma unroll
        for (int s = 0; s < NT1; s++)
            w[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW2, o2, (16 * t * h1 + 16 * s) * 4, 0));
        const f32x4 w3 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW3, nn < out_dim ? o3 : 0x7ffffff0, 16 * t * 4, 0));   // rows >= out_dim do not exist
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NT1; s++) {
            if (s & 1) acc1 = mfma4(w[s], h1r[s], acc1);
            else acc0 = mfma4(w[s], h1r[s], acc0);
        }
        const f32x4 hq = bias_relu(acc0 + acc1, b2, t * 16 + 4 * q, h2);
        if (h2_out && row_ok) *(f32x4*)(h2_out + (long)row * h2 + t * 16 + 4 * q) = hq;
        acc3 = mfma4(w3, hq, acc3);
    }
#undef KS_LDF
    if (q == 0 && row_ok) {
        const float z[4] = {acc3.x, acc3.y, acc3.z, acc3.w};
        for (int i = 0; i < out_dim; i++) {
            float y = z[i] + b3[i];
            if (act == KR_ACT_SIGMOID) y = scale / (1.f + __expf(-y));
            out[(long)row * out_dim + i] = y;
        }
    }
}

// ---- the LDS-free network with the tiles of a layer SPLIT over NWS waves of one workgroup (still 16 batch rows per workgroup).
// One wave per 16 rows is a serial chain of ~1400 MFMAs fed by ~1400 weight loads: 87 us whatever the batch, and the learner's
// five forward passes are more than half of an update that - early in training - is what an env-step waits for.  Here wave w
// computes the layer-1 tiles t = w (mod NWS), the waves exchange their output quads through global memory (the caller's
// h1_out, or scratch; L2 resident, read back with glc loads) across a workgroup barrier - no LDS, so the kernel still runs
// beside the stepping kernel - then wave w computes the layer-2 tiles t = w (mod NWS) and its share of layer 3, whose partial
// sums meet in scratch (summed in wave order: deterministic).  Per output element the layer-1 / layer-2 fma chains are those
// of k_mlp3_wave; only layer 3's sum is associated differently.
template <int NT1, int NT2, int NWS>
__global__ __launch_bounds__(64 * NWS) __attribute__((amdgpu_waves_per_eu(KS_SPLIT_WAVES_PER_EU, KS_SPLIT_WAVES_PER_EU))) void k_mlp3_split(
    int n, int in_a, int in_b, int h1, int h2, int out_dim, const float* __restrict__ xa, int lda, const float* __restrict__ xb, int ldb,
    const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ W3, const float* __restrict__ b3, int act, float scale, float* __restrict__ out, float* __restrict__ h1buf,
    int h1_rows, float* __restrict__ h2_out, float* __restrict__ partial) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nn = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * ROWS + nn;
    const bool row_ok = row < n;
    const int in_dim = in_a + in_b;
    const auto rW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W1), 0, h1 * in_dim * 4, 0x00020000);
    const auto rW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W2), 0, h2 * h1 * 4, 0x00020000);
    const auto rW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W3), 0, out_dim * h2 * 4, 0x00020000);
    const auto rXa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xa), 0, ((n - 1) * lda + in_a) * 4, 0x00020000);
    const auto rXb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb ? xb : xa), 0, xb ? ((n - 1) * ldb + in_b) * 4 : 0, 0x00020000);
    const auto rH1 = __builtin_amdgcn_make_buffer_rsrc(h1buf, 0, h1_rows * h1 * 4, 0x00020000);
#define KS_LDF(rsrc, voff, soff) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0))
    constexpr int OOR = 0x7ffffff0;
    {
        f32x4 bx[KS_IN_MAX];
        const int oa = row * lda * 4 + 16 * q, ob = (row * ldb + 4 * q - in_a) * 4;
#pragma unroll
        for (int s = 0; s < KS_IN_MAX; s++) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = 16 * s + 4 * q + j;
                const float fa = KS_LDF(rXa, (row_ok && k < in_a) ? oa : OOR, (16 * s + j) * 4);
                const float fb = KS_LDF(rXb, (row_ok && k >= in_a && k < in_dim) ? ob + (16 * s + j) * 4 : OOR, 0);
                v[j] = k < in_a ? fa : fb;
            }
            bx[s] = f32x4{v[0], v[1], v[2], v[3]};
        }
        const int o1 = (nn * in_dim + 4 * q) * 4;
#pragma unroll 1
        for (int t = wave; t < NT1; t += NWS) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS_IN_MAX; s++) {
                const int so = (16 * t * in_dim + 16 * s) * 4, k0 = 16 * s + 4 * q;
                const f32x4 w = {KS_LDF(rW1, k0 < in_dim ? o1 : OOR, so), KS_LDF(rW1, k0 + 1 < in_dim ? o1 : OOR, so + 4),
                                 KS_LDF(rW1, k0 + 2 < in_dim ? o1 : OOR, so + 8), KS_LDF(rW1, k0 + 3 < in_dim ? o1 : OOR, so + 12)};
                if (s & 1) acc1 = mfma4(w, bx[s], acc1);
                else acc0 = mfma4(w, bx[s], acc0);
            }
            const f32x4 hq = bias_relu(acc0 + acc1, b1, t * 16 + 4 * q, h1);
            if (row < h1_rows) *(f32x4*)(h1buf + (long)row * h1 + t * 16 + 4 * q) = hq;
        }
    }
    __threadfence_block();
    __syncthreads();
    // every wave takes the whole layer-1 output of its 16 rows as B operands (glc: written by the other waves of this workgroup)
    f32x4 h1r[NT1];
    {
        const int oh = row < h1_rows ? (row * h1 + 4 * q) * 4 : OOR;
#pragma unroll
        for (int s = 0; s < NT1; s++) h1r[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rH1, oh, 16 * s * 4, 1));
    }
    f32x4 acc3 = {0.f, 0.f, 0.f, 0.f};
    const int o2 = (nn * h1 + 4 * q) * 4;
    const int o3 = (nn * h2 + 4 * q) * 4;
#pragma unroll 1
    for (int t = wave; t < NT2; t += NWS) {
        f32x4 w[NT1];
#pragma unroll
        for (int s = 0; s < NT1; s++)
            w[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW2, o2, (16 * t * h1 + 16 * s) * 4, 0));
        const f32x4 w3 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW3, nn < out_dim ? o3 : OOR, 16 * t * 4, 0));
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NT1; s++) {
            if (s & 1) acc1 = mfma4(w[s], h1r[s], acc1);
            else acc0 = mfma4(w[s], h1r[s], acc0);
        }
        const f32x4 hq = bias_relu(acc0 + acc1, b2, t * 16 + 4 * q, h2);
        if (h2_out && row_ok) *(f32x4*)(h2_out + (long)row * h2 + t * 16 + 4 * q) = hq;
        acc3 = mfma4(w3, hq, acc3);
    }
#undef KS_LDF
    // layer 3: the waves' partial sums (lanes q == 0 hold the <= 4 outputs of row nn) meet in scratch
    float* pw = partial + ((long)blockIdx.x * NWS + wave) * 64;
    if (q == 0) *(f32x4*)(pw + 4 * nn) = acc3;
    __threadfence_block();
    __syncthreads();
    if (wave == 0 && q == 0 && row_ok) {
        const auto rP = __builtin_amdgcn_make_buffer_rsrc(partial + (long)blockIdx.x * NWS * 64, 0, NWS * 64 * 4, 0x00020000);
        f32x4 z4 = acc3;
#pragma unroll
        for (int w = 1; w < NWS; w++) {
            const f32x4 p = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rP, (w * 64 + 4 * nn) * 4, 0, 1));
            z4 += p;
        }
        const float z[4] = {z4.x, z4.y, z4.z, z4.w};
        for (int i = 0; i < out_dim; i++) {
            float y = z[i] + b3[i];
            if (act == KR_ACT_SIGMOID) y = scale / (1.f + __expf(-y));
            out[(long)row * out_dim + i] = y;
        }
    }
}

// ---- backward of the same network, also without LDS (one wave = 16 batch rows).
// Data gradients: with dz3 = dLoss/d(output pre-activation) [n, out_dim],
//     dz2 = (dz3 W3) * [h2 > 0],   dz1 = (dz2 W2) * [h1 > 0],   dx = dz1 W1[:, col0 : col0 + ncol]   (optional)
// again in the transposed orientation: dh^T = W^T dz^T, A = W^T (16 features of the layer below x 4 k), B = dz^T
// (4 k x 16 rows); the masked output quads are the next product's B operands, exactly as in the forward kernel.
// dx (<= 4 columns: the action inputs of the critic, DDPGfD.py:345-349) can be followed in the epilogue by the
// backward of  a = scale * sigmoid(z):  dz = dx * a (1 - a / scale)  (kr_sigmoid_scale_backward), which makes it the
// dz3 of the actor.  dz2_out / dz1_out may be NULL when only dx is wanted.
template <int NT1, int NT2>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_mlp3_bwd_wave(
    int n, int in_dim, int h1, int h2, int out_dim, const float* __restrict__ dz3, const float* __restrict__ W3, const float* __restrict__ h2a,
    const float* __restrict__ W2, const float* __restrict__ h1a, float* __restrict__ dz2_out, float* __restrict__ dz1_out,
    const float* __restrict__ W1, int col0, int ncol, const float* __restrict__ act_out, float scale, float* __restrict__ dx_out) {
    const int lane = threadIdx.x & 63, nn = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * ROWS + nn;
    const bool row_ok = row < n;
    // B operand of the first product: dz3^T, k = output index = q
    const float b3 = (row_ok && q < out_dim) ? dz3[(long)row * out_dim + q] : 0.f;
    f32x4 dz2r[NT2];
#pragma unroll
    for (int t = 0; t < NT2; t++) {
        const int f = t * 16 + nn;                                          // A: W3^T[f][k = q] = W3[q][f]
        const float a3 = (q < out_dim && f < h2) ? W3[(long)q * h2 + f] : 0.f;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc, 0, 0, 0);
        const int f4 = t * 16 + 4 * q;
        f32x4 hv = {0.f, 0.f, 0.f, 0.f};
        if (row_ok && f4 < h2) hv = *(const f32x4*)(h2a + (long)row * h2 + f4);
        f32x4 dz;
        dz.x = hv.x > 0.f ? acc.x : 0.f; dz.y = hv.y > 0.f ? acc.y : 0.f; dz.z = hv.z > 0.f ? acc.z : 0.f; dz.w = hv.w > 0.f ? acc.w : 0.f;
        dz2r[t] = dz;
        if (dz2_out && row_ok && f4 < h2) *(f32x4*)(dz2_out + (long)row * h2 + f4) = dz;
    }
    f32x4 accx = {0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W2), 0, h1 * h2 * 4, 0x00020000);
#pragma unroll 1
    for (int t = 0; t < NT1; t++) {
        // A: W2^T[f][k] = W2[k][f], k = 16 s + 4 q + j (rows of W2, stride h1), f = 16 t + nn
        const int f = t * 16 + nn;
        // (host checks h1 % 16 == h2 % 16 == 0: every row / column of the tile exists).  Buffer loads: ONE 32-bit lane
        // offset for the whole tile, the row steps (16 s + j) * h1 are wave-uniform and go in the scalar offset - flat
        // loads would hold a 64-bit address per load in flight and spill at this register budget.
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const int voff = (4 * q * h1 + f) * 4;
        constexpr int HALF = (NT2 + 1) / 2;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            f32x4 w[HALF];
#pragma unroll
            for (int u = 0; u < HALF; u++) {
                const int s = half * HALF + u;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (s < NT2) {
                    v.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW2, voff, (16 * s + 0) * h1 * 4, 0));
                    v.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW2, voff, (16 * s + 1) * h1 * 4, 0));
                    v.z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW2, voff, (16 * s + 2) * h1 * 4, 0));
                    v.w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW2, voff, (16 * s + 3) * h1 * 4, 0));
                }
                w[u] = v;
            }
#pragma unroll
            for (int u = 0; u < HALF; u++) {
                const int s = half * HALF + u;
                if (s < NT2) {
                    if (s & 1) acc1 = mfma4(w[u], dz2r[s], acc1);
                    else acc0 = mfma4(w[u], dz2r[s], acc0);
                }
            }
        }
        const f32x4 acc = acc0 + acc1;
        const int f4 = t * 16 + 4 * q;
        f32x4 hv = {0.f, 0.f, 0.f, 0.f};
        if (row_ok && f4 < h1) hv = *(const f32x4*)(h1a + (long)row * h1 + f4);
        f32x4 dz;
        dz.x = hv.x > 0.f ? acc.x : 0.f; dz.y = hv.y > 0.f ? acc.y : 0.f; dz.z = hv.z > 0.f ? acc.z : 0.f; dz.w = hv.w > 0.f ? acc.w : 0.f;
        if (dz1_out && row_ok && f4 < h1) *(f32x4*)(dz1_out + (long)row * h1 + f4) = dz;
        if (dx_out) {
            // A: W1[:, col0 + m]^T: [m][k] = W1[k][col0 + m], k = 16 t + 4 q + j (rows of W1, stride in_dim), m = nn < ncol
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (nn < ncol) {
                const int k = 16 * t + 4 * q;
                const float* p = W1 + (long)k * in_dim + col0 + nn;
                if (k < h1) v.x = p[0];
                if (k + 1 < h1) v.y = p[in_dim];
                if (k + 2 < h1) v.z = p[2 * (long)in_dim];
                if (k + 3 < h1) v.w = p[3 * (long)in_dim];
            }
            accx = mfma4(v, dz, accx);
        }
    }
    if (dx_out && q == 0 && row_ok) {
        const float g[4] = {accx.x, accx.y, accx.z, accx.w};
        for (int i = 0; i < ncol; i++) {
            float v = g[i];
            if (act_out) { const float a = act_out[(long)row * ncol + i]; v *= a * (1.f - a / scale); }
            dx_out[(long)row * ncol + i] = v;
        }
    }
}

// ... and with the tiles of dz1 split over NWS waves of a workgroup (as k_mlp3_split): every wave computes dz2 itself (one MFMA per
// tile), wave w the dz1 tiles t = w (mod NWS); the partial sums of dx meet in scratch, summed in wave order.
template <int NT1, int NT2, int NWS>
__global__ __launch_bounds__(64 * NWS) __attribute__((amdgpu_waves_per_eu(KS_SPLIT_WAVES_PER_EU, KS_SPLIT_WAVES_PER_EU))) void k_mlp3_bwd_split(
    int n, int in_dim, int h1, int h2, int out_dim, const float* __restrict__ dz3, const float* __restrict__ W3, const float* __restrict__ h2a,
    const float* __restrict__ W2, const float* __restrict__ h1a, float* __restrict__ dz2_out, float* __restrict__ dz1_out,
    const float* __restrict__ W1, int col0, int ncol, const float* __restrict__ act_out, float scale, float* __restrict__ dx_out,
    float* __restrict__ partial) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nn = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * ROWS + nn;
    const bool row_ok = row < n;
    // B operand of the first product: dz3^T, k = output index = q
    const float b3 = (row_ok && q < out_dim) ? dz3[(long)row * out_dim + q] : 0.f;
    f32x4 dz2r[NT2];
#pragma unroll
    for (int t = 0; t < NT2; t++) {
        const int f = t * 16 + nn;                                          // A: W3^T[f][k = q] = W3[q][f]
        const float a3 = (q < out_dim && f < h2) ? W3[(long)q * h2 + f] : 0.f;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc, 0, 0, 0);
        const int f4 = t * 16 + 4 * q;
        f32x4 hv = {0.f, 0.f, 0.f, 0.f};
        if (row_ok && f4 < h2) hv = *(const f32x4*)(h2a + (long)row * h2 + f4);
        f32x4 dz;
        dz.x = hv.x > 0.f ? acc.x : 0.f; dz.y = hv.y > 0.f ? acc.y : 0.f; dz.z = hv.z > 0.f ? acc.z : 0.f; dz.w = hv.w > 0.f ? acc.w : 0.f;
        dz2r[t] = dz;
        if (dz2_out && row_ok && f4 < h2 && t % NWS == wave) *(f32x4*)(dz2_out + (long)row * h2 + f4) = dz;
    }
    f32x4 accx = {0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W2), 0, h1 * h2 * 4, 0x00020000);
#pragma unroll 1
    for (int t = wave; t < NT1; t += NWS) {
        // A: W2^T[f][k] = W2[k][f], k = 16 s + 4 q + j (rows of W2, stride h1), f = 16 t + nn
        const int f = t * 16 + nn;
        // (host checks h1 % 16 == h2 % 16 == 0: every row / column of the tile exists).  Buffer loads: ONE 32-bit lane
        // offset for the whole tile, the row steps (16 s + j) * h1 are wave-uniform and go in the scalar offset - flat
        // loads would hold a 64-bit address per load in flight and spill at this register budget.
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const int voff = (4 * q * h1 + f) * 4;
        constexpr int HALF = (NT2 + 1) / 2;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            f32x4 w[HALF];
#pragma unroll
            for (int u = 0; u < HALF; u++) {
                const int s = half * HALF + u;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (s < NT2) {
                    v.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW2, voff, (16 * s + 0) * h1 * 4, 0));
                    v.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW2, voff, (16 * s + 1) * h1 * 4, 0));
                    v.z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW2, voff, (16 * s + 2) * h1 * 4, 0));
                    v.w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW2, voff, (16 * s + 3) * h1 * 4, 0));
                }
                w[u] = v;
            }
#pragma unroll
            for (int u = 0; u < HALF; u++) {
                const int s = half * HALF + u;
                if (s < NT2) {
                    if (s & 1) acc1 = mfma4(w[u], dz2r[s], acc1);
                    else acc0 = mfma4(w[u], dz2r[s], acc0);
                }
            }
        }
        const f32x4 acc = acc0 + acc1;
        const int f4 = t * 16 + 4 * q;
        f32x4 hv = {0.f, 0.f, 0.f, 0.f};
        if (row_ok && f4 < h1) hv = *(const f32x4*)(h1a + (long)row * h1 + f4);
        f32x4 dz;
        dz.x = hv.x > 0.f ? acc.x : 0.f; dz.y = hv.y > 0.f ? acc.y : 0.f; dz.z = hv.z > 0.f ? acc.z : 0.f; dz.w = hv.w > 0.f ? acc.w : 0.f;
        if (dz1_out && row_ok && f4 < h1) *(f32x4*)(dz1_out + (long)row * h1 + f4) = dz;
        if (dx_out) {
            // A: W1[:, col0 + m]^T: [m][k] = W1[k][col0 + m], k = 16 t + 4 q + j (rows of W1, stride in_dim), m = nn < ncol
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (nn < ncol) {
                const int k = 16 * t + 4 * q;
                const float* p = W1 + (long)k * in_dim + col0 + nn;
                if (k < h1) v.x = p[0];
                if (k + 1 < h1) v.y = p[in_dim];
                if (k + 2 < h1) v.z = p[2 * (long)in_dim];
                if (k + 3 < h1) v.w = p[3 * (long)in_dim];
            }
            accx = mfma4(v, dz, accx);
        }
    }
    if (dx_out) {
        float* pw = partial + ((long)blockIdx.x * NWS + wave) * 64;
        if (q == 0) *(f32x4*)(pw + 4 * nn) = accx;
        __threadfence_block();
        __syncthreads();
        if (wave == 0 && q == 0) {
            const auto rP = __builtin_amdgcn_make_buffer_rsrc(partial + (long)blockIdx.x * NWS * 64, 0, NWS * 64 * 4, 0x00020000);
#pragma unroll
            for (int w = 1; w < NWS; w++) accx += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rP, (w * 64 + 4 * nn) * 4, 0, 1));
        }
    }
    if (dx_out && wave == 0 && q == 0 && row_ok) {
        const float g[4] = {accx.x, accx.y, accx.z, accx.w};
        for (int i = 0; i < ncol; i++) {
            float v = g[i];
            if (act_out) { const float a = act_out[(long)row * ncol + i]; v *= a * (1.f - a / scale); }
            dx_out[(long)row * ncol + i] = v;
        }
    }
}

// Weight gradients  dW[M][N] = dz^T h  (dz [n][M], h = [ha | hb] [n][N]) and the bias gradient  db[M] = column sums of dz,
// without LDS: a wave owns one 16-row tile of dW and TN 16-column tiles, and one chunk of the batch rows; A = dz^T
// (lane: feature m, row k), B = h (lane: row k, column).  The chunk partials go to a workspace [chunk][M * N + M] that
// k_wgrad_reduce sums in chunk order (deterministic, unlike atomics).
constexpr int WG_TN = 4;
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_wgrad_wave(
    int n, int M, int Na, int Nb, int rows_per_chunk, const float* __restrict__ dz, const float* __restrict__ ha, int lda,
    const float* __restrict__ hb, int ldb, float* __restrict__ ws) {
    const int lane = threadIdx.x & 63, nn = lane & 15, q = lane >> 4;
    const int N = Na + Nb, n_blocks = (N + 16 * WG_TN - 1) / (16 * WG_TN);
    const int mt = blockIdx.x / n_blocks, nb = blockIdx.x % n_blocks, chunk = blockIdx.y;
    const int r0 = chunk * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
    const int m = mt * 16 + nn;
    // Raw buffer loads (as in k_mlp3_wave): per matrix ONE lane offset (row q of a 4-row group, this lane's column), the row
    // advance in the wave-uniform scalar offset; an element that does not exist gets an out-of-range lane offset (-> 0).
    // The flat-load version spent ~10 VALU + ~20 SALU instructions per MFMA on addresses and bounds branches.
    constexpr int OOR = 0x7ffffff0;
    const auto rZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dz), 0, n * M * 4, 0x00020000);
    const auto rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ha), 0, ((n - 1) * lda + Na) * 4, 0x00020000);
    const auto rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hb ? hb : ha), 0, hb ? ((n - 1) * ldb + Nb) * 4 : 0, 0x00020000);
#define KS_LDF(rsrc, voff, soff) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0))
    const int vz = m < M ? (q * M + m) * 4 : OOR;
    int va[WG_TN], vb[WG_TN];
#pragma unroll
    for (int u = 0; u < WG_TN; u++) {
        const int c = (nb * WG_TN + u) * 16 + nn;
        va[u] = c < Na ? (q * lda + c) * 4 : OOR;
        vb[u] = (c >= Na && c < N) ? (q * ldb + c - Na) * 4 : OOR;
    }
    f32x4 acc[WG_TN];
#pragma unroll
    for (int u = 0; u < WG_TN; u++) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    float colsum = 0.f;
    // four k-steps (16 rows) per round, all loads first; only the last round of a chunk can hold rows that do not exist
    auto round = [&](int k0, auto guard, auto hasb) {
        constexpr bool GUARD = decltype(guard)::value, HASB = decltype(hasb)::value;
        float a[4], b[4][WG_TN];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int rj = k0 + 4 * j;                                 // wave-uniform; this lane's row is rj + q
            const bool ok = !GUARD || rj + q < r1;
            a[j] = KS_LDF(rZ, ok ? vz : OOR, rj * M * 4);
#pragma unroll
            for (int u = 0; u < WG_TN; u++) {
                float v = KS_LDF(rA, ok ? va[u] : OOR, rj * lda * 4);
                if (HASB) v += KS_LDF(rB, ok ? vb[u] : OOR, rj * ldb * 4);      // at most one of the two exists
                b[j][u] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            colsum += a[j];
#pragma unroll
            for (int u = 0; u < WG_TN; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j][u], acc[u], 0, 0, 0);
        }
    };
    const int full_end = r0 + ((r1 - r0) >> 4 << 4);
    if (Nb > 0) {
        for (int k0 = r0; k0 < full_end; k0 += 16) round(k0, std::false_type{}, std::true_type{});
        if (full_end < r1) round(full_end, std::true_type{}, std::true_type{});
    } else {
        for (int k0 = r0; k0 < full_end; k0 += 16) round(k0, std::false_type{}, std::false_type{});
        if (full_end < r1) round(full_end, std::true_type{}, std::false_type{});
    }
#undef KS_LDF
    float* out = ws + (long)chunk * ((long)M * N + M);
#pragma unroll
    for (int u = 0; u < WG_TN; u++) {
        const int c = (nb * WG_TN + u) * 16 + nn;
        const float v[4] = {acc[u].x, acc[u].y, acc[u].z, acc[u].w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int mr = mt * 16 + 4 * q + i;
            if (mr < M && c < N) out[(long)mr * N + c] = v[i];
        }
    }
    if (nb == 0) {
        colsum += __shfl_xor(colsum, 16);
        colsum += __shfl_xor(colsum, 32);
        if (q == 0 && m < M) out[(long)M * N + m] = colsum;
    }
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(long count_w, long count_b, int chunks, const float* __restrict__ ws, float* __restrict__ dW,
                                                      float* __restrict__ db) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x, total = count_w + count_b;
    if (i >= total) return;
    float s = 0.f;
    for (int c = 0; c < chunks; c++) s += ws[(long)c * total + i];
    if (i < count_w) dW[i] = s;
    else db[i - count_w] = s;
}

template <int NT1, int NT2, bool SEL>
int launch(int n, int in_a, int in_b, int h1, int h2, int out_dim, const float* xa, int lda, const float* xb, int ldb, const float* W1,
           const float* b1, const float* W2, const float* b2, const float* W3, const float* b3, int act, float scale, float* out,
           float* h1_out, float* h2_out, const SelectArgs& sel, hipStream_t s) {
    const bool vec = (h1 % 4 == 0) && (h2 % 4 == 0) && ((uintptr_t)W2 % 16 == 0) && ((uintptr_t)W3 % 16 == 0);
    const dim3 grid((n + ROWS - 1) / ROWS), block(64 * NW);
    if (vec)
        hipLaunchKernelGGL((k_mlp3<NT1, NT2, true, SEL>), grid, block, 0, s, n, in_a, in_b, h1, h2, out_dim, xa, lda, xb, ldb, W1, b1, W2, b2, W3, b3,
                           act, scale, out, h1_out, h2_out, sel);
    else
        hipLaunchKernelGGL((k_mlp3<NT1, NT2, false, SEL>), grid, block, 0, s, n, in_a, in_b, h1, h2, out_dim, xa, lda, xb, ldb, W1, b1, W2, b2, W3, b3,
                           act, scale, out, h1_out, h2_out, sel);
    return hipGetLastError() == hipSuccess ? KS_OK : KS_ERR_HIP;
}


template <bool SEL>
int dispatch(int n, int in_a, int in_b, int h1, int h2, int out_dim, const float* xa, int lda, const float* xb, int ldb, const float* W1,
             const float* b1, const float* W2, const float* b2, const float* W3, const float* b3, int act, float scale, float* out,
             float* h1_out, float* h2_out, const SelectArgs& sel, hipStream_t s) {
    const int nt1 = (h1 + 15) / 16, nt2 = (h2 + 15) / 16;
#define KR_MLP_CASE(A, B) \
    if (nt1 == A && nt2 == B) return launch<A, B, SEL>(n, in_a, in_b, h1, h2, out_dim, xa, lda, xb, ldb, W1, b1, W2, b2, W3, b3, act, scale, out, h1_out, h2_out, sel, s);
    KR_MLP_CASE(16, 16)      // 256-256 (BASELINE)
    KR_MLP_CASE(25, 19)      // 400-300 (reference, DDPGfD.py:19-23)
    KR_MLP_CASE(8, 8)        // 128-128
    KR_MLP_CASE(4, 4)        // 64-64 (tests)
#undef KR_MLP_CASE
    return KS_ERR_INVALID;   // other widths: the caller keeps its GEMM path
}

}  // namespace

extern "C" {

int kr_mlp3_forward(int32_t n, int32_t in_a, int32_t in_b, int32_t h1, int32_t h2, int32_t out_dim, const float* xa, int32_t lda,
                    const float* xb, int32_t ldb, const float* W1, const float* b1, const float* W2, const float* b2,
                    const float* W3, const float* b3, int32_t act, float scale, float* out, float* h1_out, float* h2_out, void* stream) {
    if (n <= 0) return KS_OK;
    if (((h1_out && (h1 % 4 || (uintptr_t)h1_out % 16)) || (h2_out && (h2 % 4 || (uintptr_t)h2_out % 16)))) return KS_ERR_INVALID;
    if (!xa || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !out || in_a <= 0 || in_b < 0 || (in_b > 0 && !xb)) return KS_ERR_INVALID;
    if (in_a + in_b > 16 * KS_IN_MAX || out_dim < 1 || out_dim > 4 || h1 < 1 || h2 < 1) return KS_ERR_INVALID;
    if (act != KR_ACT_NONE && act != KR_ACT_SIGMOID) return KS_ERR_INVALID;
    return dispatch<false>(n, in_a, in_b, h1, h2, out_dim, xa, lda, xb, ldb, W1, b1, W2, b2, W3, b3, act, scale, out, h1_out, h2_out, SelectArgs{}, (hipStream_t)stream);
}

int kr_mlp3_forward_shadow(int32_t n, int32_t in_a, int32_t in_b, int32_t h1, int32_t h2, int32_t out_dim, const float* xa, int32_t lda,
                           const float* xb, int32_t ldb, const float* W1, const float* b1, const float* W2, const float* b2,
                           const float* W3, const float* b3, int32_t act, float scale, float* out, float* h1_out, float* h2_out, void* stream) {
    if (n <= 0) return KS_OK;
    if (((h1_out && (h1 % 4 || (uintptr_t)h1_out % 16)) || (h2_out && (h2 % 4 || (uintptr_t)h2_out % 16)))) return KS_ERR_INVALID;
    if (!xa || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !out || in_a <= 0 || in_b < 0 || (in_b > 0 && !xb)) return KS_ERR_INVALID;
    if (in_a + in_b > 16 * KS_IN_MAX || out_dim < 1 || out_dim > 4 || h1 < 1 || h2 < 1) return KS_ERR_INVALID;
    if (act != KR_ACT_NONE && act != KR_ACT_SIGMOID) return KS_ERR_INVALID;
    const int nt1 = (h1 + 15) / 16, nt2 = (h2 + 15) / 16;
    if (h1 % 16 || h2 % 16 || (uintptr_t)W2 % 16 || (uintptr_t)W3 % 16) return KS_ERR_INVALID;      // 16-byte buffer loads, whole tiles
    const bool vec = true;
    const dim3 grid((n + ROWS - 1) / ROWS), block(64);
    hipStream_t s = (hipStream_t)stream;
#define KR_WAVE_CASE(A, B)                                                                                                                       \
    if (nt1 == A && nt2 == B) {                                                                                                                  \
        if (vec) hipLaunchKernelGGL((k_mlp3_wave<A, B, true>), grid, block, 0, s, n, in_a, in_b, h1, h2, out_dim, xa, lda, xb, ldb, W1, b1, W2, b2, \
                                    W3, b3, act, scale, out, h1_out, h2_out);                                                                    \
        else hipLaunchKernelGGL((k_mlp3_wave<A, B, false>), grid, block, 0, s, n, in_a, in_b, h1, h2, out_dim, xa, lda, xb, ldb, W1, b1, W2, b2,   \
                                W3, b3, act, scale, out, h1_out, h2_out);                                                                        \
        return hipGetLastError() == hipSuccess ? KS_OK : KS_ERR_HIP;                                                                             \
    }
    KR_WAVE_CASE(16, 16)     // 256-256 (BASELINE); wider first layers do not fit the register budget
    KR_WAVE_CASE(8, 8)
    KR_WAVE_CASE(4, 4)
#undef KR_WAVE_CASE
    return KS_ERR_INVALID;
}

int kr_mlp3_forward_split(int32_t n, int32_t in_a, int32_t in_b, int32_t h1, int32_t h2, int32_t out_dim, const float* xa, int32_t lda,
                          const float* xb, int32_t ldb, const float* W1, const float* b1, const float* W2, const float* b2,
                          const float* W3, const float* b3, int32_t act, float scale, float* out, float* h1_out, float* h2_out,
                          float* scratch, int64_t scratch_floats, int32_t waves, void* stream) {
    if (n <= 0) return KS_OK;
    if (((h1_out && (h1 % 4 || (uintptr_t)h1_out % 16)) || (h2_out && (h2 % 4 || (uintptr_t)h2_out % 16)))) return KS_ERR_INVALID;
    if (!xa || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !out || in_a <= 0 || in_b < 0 || (in_b > 0 && !xb)) return KS_ERR_INVALID;
    if (in_a + in_b > 16 * KS_IN_MAX || out_dim < 1 || out_dim > 4 || h1 < 1 || h2 < 1) return KS_ERR_INVALID;
    if (act != KR_ACT_NONE && act != KR_ACT_SIGMOID) return KS_ERR_INVALID;
    if (h1 % 16 || h2 % 16 || (uintptr_t)W2 % 16 || (uintptr_t)W3 % 16 || (waves != 2 && waves != 4)) return KS_ERR_INVALID;
    const int nt1 = h1 / 16, nt2 = h2 / 16, blocks = (n + ROWS - 1) / ROWS;
    // scratch: [the layer-1 exchange, blocks x 16 rows x h1, unless the caller keeps h1 itself][blocks x waves x 64 partial sums]
    const int64_t need = (h1_out ? 0 : (int64_t)blocks * ROWS * h1) + (int64_t)blocks * waves * 64;
    if (!scratch || (uintptr_t)scratch % 16 || scratch_floats < need) return KS_ERR_INVALID;
    float* h1buf = h1_out ? h1_out : scratch;
    const int h1_rows = h1_out ? n : blocks * ROWS;
    float* partial = scratch + (h1_out ? 0 : (int64_t)blocks * ROWS * h1);
    const dim3 grid(blocks);
    hipStream_t s = (hipStream_t)stream;
#define KR_SPLIT_CASE(A, B, NWS)                                                                                                               \
    if (nt1 == A && nt2 == B && waves == NWS) {                                                                                                \
        hipLaunchKernelGGL((k_mlp3_split<A, B, NWS>), grid, dim3(64 * NWS), 0, s, n, in_a, in_b, h1, h2, out_dim, xa, lda, xb, ldb, W1, b1, W2, b2, \
                           W3, b3, act, scale, out, h1buf, h1_rows, h2_out, partial);                                                          \
        return hipGetLastError() == hipSuccess ? KS_OK : KS_ERR_HIP;                                                                           \
    }
    KR_SPLIT_CASE(16, 16, 4) KR_SPLIT_CASE(16, 16, 2)
    KR_SPLIT_CASE(8, 8, 4) KR_SPLIT_CASE(8, 8, 2)
    KR_SPLIT_CASE(4, 4, 4) KR_SPLIT_CASE(4, 4, 2)
#undef KR_SPLIT_CASE
    return KS_ERR_INVALID;
}

int kr_mlp3_backward_shadow(int32_t n, int32_t in_dim, int32_t h1, int32_t h2, int32_t out_dim, const float* dz3, const float* W3,
                            const float* h2a, const float* W2, const float* h1a, float* dz2_out, float* dz1_out, const float* W1, int32_t col0,
                            int32_t ncol, const float* act_out, float scale, float* dx_out, void* stream) {
    if (n <= 0) return KS_OK;
    if (!dz3 || !W3 || !h2a || !W2 || !h1a || out_dim < 1 || out_dim > 4 || h1 < 16 || h2 < 16 || h1 % 16 || h2 % 16) return KS_ERR_INVALID;
    if (dx_out && (!W1 || ncol < 1 || ncol > 4 || col0 < 0 || col0 + ncol > in_dim)) return KS_ERR_INVALID;
    if ((uintptr_t)h1a % 16 || (uintptr_t)h2a % 16 || (dz1_out && (uintptr_t)dz1_out % 16) || (dz2_out && (uintptr_t)dz2_out % 16)) return KS_ERR_INVALID;
    const int nt1 = (h1 + 15) / 16, nt2 = (h2 + 15) / 16;
    const dim3 grid((n + ROWS - 1) / ROWS), block(64);
    hipStream_t s = (hipStream_t)stream;
#define KR_BWD_CASE(A, B)                                                                                                                    \
    if (nt1 == A && nt2 == B) {                                                                                                              \
        hipLaunchKernelGGL((k_mlp3_bwd_wave<A, B>), grid, block, 0, s, n, in_dim, h1, h2, out_dim, dz3, W3, h2a, W2, h1a, dz2_out, dz1_out, W1, \
                           col0, ncol, act_out, scale, dx_out);                                                                              \
        return hipGetLastError() == hipSuccess ? KS_OK : KS_ERR_HIP;                                                                         \
    }
    KR_BWD_CASE(16, 16)
    KR_BWD_CASE(8, 8)
    KR_BWD_CASE(4, 4)
#undef KR_BWD_CASE
    return KS_ERR_INVALID;
}

int kr_mlp3_backward_split(int32_t n, int32_t in_dim, int32_t h1, int32_t h2, int32_t out_dim, const float* dz3, const float* W3,
                            const float* h2a, const float* W2, const float* h1a, float* dz2_out, float* dz1_out, const float* W1, int32_t col0,
                            int32_t ncol, const float* act_out, float scale, float* dx_out, float* scratch, int64_t scratch_floats, int32_t waves, void* stream) {
    if (n <= 0) return KS_OK;
    if (!dz3 || !W3 || !h2a || !W2 || !h1a || out_dim < 1 || out_dim > 4 || h1 < 16 || h2 < 16 || h1 % 16 || h2 % 16) return KS_ERR_INVALID;
    if (dx_out && (!W1 || ncol < 1 || ncol > 4 || col0 < 0 || col0 + ncol > in_dim)) return KS_ERR_INVALID;
    if ((uintptr_t)h1a % 16 || (uintptr_t)h2a % 16 || (dz1_out && (uintptr_t)dz1_out % 16) || (dz2_out && (uintptr_t)dz2_out % 16)) return KS_ERR_INVALID;
    const int nt1 = (h1 + 15) / 16, nt2 = (h2 + 15) / 16;
    if ((waves != 2 && waves != 4) || (dx_out && (!scratch || (uintptr_t)scratch % 16 || scratch_floats < (int64_t)((n + ROWS - 1) / ROWS) * waves * 64)))
        return KS_ERR_INVALID;
    const dim3 grid((n + ROWS - 1) / ROWS);
    hipStream_t s = (hipStream_t)stream;
#define KR_BWDS_CASE(A, B, NWS)                                                                                                              \
    if (nt1 == A && nt2 == B && waves == NWS) {                                                                                              \
        hipLaunchKernelGGL((k_mlp3_bwd_split<A, B, NWS>), grid, dim3(64 * NWS), 0, s, n, in_dim, h1, h2, out_dim, dz3, W3, h2a, W2, h1a, dz2_out, \
                           dz1_out, W1, col0, ncol, act_out, scale, dx_out, scratch);                                                        \
        return hipGetLastError() == hipSuccess ? KS_OK : KS_ERR_HIP;                                                                         \
    }
    KR_BWDS_CASE(16, 16, 4) KR_BWDS_CASE(16, 16, 2)
    KR_BWDS_CASE(8, 8, 4) KR_BWDS_CASE(8, 8, 2)
    KR_BWDS_CASE(4, 4, 4) KR_BWDS_CASE(4, 4, 2)
#undef KR_BWDS_CASE
    return KS_ERR_INVALID;
}

int kr_weight_grad_shadow(int32_t n, int32_t M, int32_t Na, int32_t Nb, const float* dz, const float* ha, int32_t lda, const float* hb, int32_t ldb,
                          int32_t chunks, float* workspace, float* dW, float* db, void* stream) {
    if (n <= 0 || M < 1 || Na < 1 || Nb < 0 || chunks < 1 || !dz || !ha || (Nb > 0 && !hb) || !workspace || !dW || !db) return KS_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int N = Na + Nb, n_blocks = (N + 16 * WG_TN - 1) / (16 * WG_TN), m_tiles = (M + 15) / 16;
    int rows_per_chunk = (n + chunks - 1) / chunks;
    rows_per_chunk = (rows_per_chunk + 15) / 16 * 16;
    hipLaunchKernelGGL(k_wgrad_wave, dim3(m_tiles * n_blocks, chunks), dim3(64), 0, s, n, M, Na, Nb, rows_per_chunk, dz, ha, lda, hb, ldb, workspace);
    const long total = (long)M * N + M;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (long)M * N, (long)M, chunks, workspace, dW, db);
    return hipGetLastError() == hipSuccess ? KS_OK : KS_ERR_HIP;
}

int kr_actor_select(int32_t n, int32_t h1, int32_t h2, const float* obs, const float* prev_obs, const uint8_t* has_prev, const int64_t* t,
                    uint8_t* ready, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3, const float* b3,
                    const float* noise, uint64_t seed, int64_t* rng_state, float sigma, float max_action, int32_t skip_steps, float* actor_out,
                    float* action, float* action_t, uint8_t* lifting, void* stream) {
    if (n <= 0) return KS_OK;
    if (!obs || !prev_obs || !has_prev || !t || !ready || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !action || !action_t || !lifting) return KS_ERR_INVALID;
    if ((noise == nullptr) == (rng_state == nullptr) || h1 < 1 || h2 < 1) return KS_ERR_INVALID;     // exactly one noise source
    SelectArgs sel{obs, prev_obs, has_prev, t, ready, noise, (unsigned long long)seed, rng_state, sigma, max_action, skip_steps, action, action_t, lifting};
    return dispatch<true>(n, KR_STATE_DIM, 0, h1, h2, KR_ACTION_DIM, obs, KR_STATE_DIM, nullptr, 0, W1, b1, W2, b2, W3, b3, KR_ACT_SIGMOID, max_action,
                          actor_out, nullptr, nullptr, sel, (hipStream_t)stream);
}

}  // extern "C"
