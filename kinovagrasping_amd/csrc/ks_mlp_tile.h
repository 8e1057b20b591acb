// ks_mlp_tile.h -- the 3-layer MLP forward of ONE 16-row tile on the matrix cores, as a device function: the body of k_mlp3
// (ks_mlp.hip: kr_mlp3_forward / kr_actor_select) and of the in-kernel actor of the free-running rollout kernel (ks_api.hip:
// k_rollout), so that both are the same k-ordered fp32 fma chains bit for bit.  See ks_mlp.hip for the layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kmlp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWS = 16;        // batch rows per workgroup (the N of the MFMA)
constexpr int KS_IN_MAX = 6;    // input k-steps: in_dim <= 96
#ifndef KS_MLP_WAVES
#define KS_MLP_WAVES 4
#endif
constexpr int NW = KS_MLP_WAVES;   // waves per workgroup

// 4 consecutive weights W[row][k .. k+3] (zero beyond the matrix): one 16-byte load when the row is 16-byte aligned
template <bool VEC> __device__ __forceinline__ f32x4 load_w4(const float* __restrict__ W, int row, int nrow, int k, int K) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row >= nrow) return v;
    const float* p = W + (long)row * K + k;
    if (VEC && k + 3 < K) return *(const f32x4*)p;
    if (k < K) v.x = p[0];
    if (k + 1 < K) v.y = p[1];
    if (k + 2 < K) v.z = p[2];
    if (k + 3 < K) v.w = p[3];
    return v;
}

__device__ __forceinline__ f32x4 mfma4(f32x4 a, f32x4 b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
    return c;
}

// bias + ReLU on an output quad (features f .. f+3, zero beyond nfeat)
__device__ __forceinline__ f32x4 bias_relu(f32x4 acc, const float* __restrict__ bias, int f, int nfeat) {
    f32x4 r;
    r.x = f < nfeat ? fmaxf(acc.x + bias[f], 0.f) : 0.f;
    r.y = f + 1 < nfeat ? fmaxf(acc.y + bias[f + 1], 0.f) : 0.f;
    r.z = f + 2 < nfeat ? fmaxf(acc.z + bias[f + 2], 0.f) : 0.f;
    r.w = f + 3 < nfeat ? fmaxf(acc.w + bias[f + 3], 0.f) : 0.f;
    return r;
}

// All three layers for the 16 rows of a workgroup of NW waves (every thread of the workgroup must call this: two barriers inside).
// `row` = this lane's batch row (lane & 15 selects it; the same in all waves), < 0: no such row.  H1 / H2 / P: workgroup-shared
// scratch, [NT1 * 4][ROWS], [NT2 * 4][ROWS], [NW][ROWS] float4.  Returns true on the lanes that hold a row's layer-3 sums z4
// (wave 0, first quarter, row valid): the caller adds b3 and applies the output activation.
template <int NT1, int NT2, bool VEC>
__device__ __forceinline__ bool mlp3_rows16(const int wave, const int lane, const long row, int in_a, int in_b, int h1, int h2, int out_dim,
                                            const float* __restrict__ xa, int lda, const float* __restrict__ xb, int ldb,
                                            const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                                            const float* __restrict__ b2, const float* __restrict__ W3, float* __restrict__ h1_out,
                                            float* __restrict__ h2_out, f32x4 (*H1)[ROWS], f32x4 (*H2)[ROWS], f32x4 (*P)[ROWS], f32x4& z4) {
    const int nn = lane & 15, q = lane >> 4;
    const bool row_ok = row >= 0;
    const int in_dim = in_a + in_b;
    // layer 1: the wave's copy of the 16 input rows as B operands (k = 16 s + 4 q + j)
    f32x4 bx[KS_IN_MAX];
#pragma unroll
    for (int s = 0; s < KS_IN_MAX; s++) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = 16 * s + 4 * q + j;
            float x = 0.f;
            if (row_ok && k < in_a) x = xa[(long)row * lda + k];
            else if (row_ok && k < in_dim) x = xb[(long)row * ldb + (k - in_a)];
            v[j] = x;
        }
        bx[s] = f32x4{v[0], v[1], v[2], v[3]};
    }
    // Weights stream from L2 (~1.5 us per dependent read at one wave per SIMD): every tile's weights are requested one
    // tile ahead of the MFMAs that use them, and the first tile of a layer while the previous layer is still computing.
    f32x4 w2[NT1];
#pragma unroll
    for (int s = 0; s < NT1; s++) w2[s] = load_w4<VEC>(W2, wave * 16 + nn, wave < NT2 ? h2 : 0, 16 * s + 4 * q, h1);
    {
        f32x4 w1[KS_IN_MAX];
#pragma unroll
        for (int s = 0; s < KS_IN_MAX; s++) w1[s] = load_w4<false>(W1, wave * 16 + nn, wave < NT1 ? h1 : 0, 16 * s + 4 * q, in_dim);
        for (int t = wave; t < NT1; t += NW) {
            f32x4 wn[KS_IN_MAX];
#pragma unroll
            for (int s = 0; s < KS_IN_MAX; s++) wn[s] = load_w4<false>(W1, (t + NW) * 16 + nn, t + NW < NT1 ? h1 : 0, 16 * s + 4 * q, in_dim);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};  // two chains: the MFMA's dependent latency is 40 cycles, its issue 32
#pragma unroll
            for (int s = 0; s < KS_IN_MAX; s++) {       // k beyond in_dim: both operands are zero
                if (s & 1) acc1 = mfma4(w1[s], bx[s], acc1);
                else acc0 = mfma4(w1[s], bx[s], acc0);
            }
            const f32x4 hq = bias_relu(acc0 + acc1, b1, t * 16 + 4 * q, h1);
            H1[t * 4 + q][nn] = hq;
            if (h1_out && row_ok && t * 16 + 4 * q < h1) *(f32x4*)(h1_out + (long)row * h1 + t * 16 + 4 * q) = hq;   // h1 % 4 == 0 (checked by the host)
#pragma unroll
            for (int s = 0; s < KS_IN_MAX; s++) w1[s] = wn[s];
        }
    }
    __syncthreads();

    // layer 2: K = h1, one k-step per tile of H1
    f32x4 w3[(NT2 + NW - 1) / NW];
#pragma unroll
    for (int j = 0; j < (NT2 + NW - 1) / NW; j++) w3[j] = load_w4<VEC>(W3, nn, out_dim, 16 * (wave + NW * j) + 4 * q, h2);
    for (int t = wave; t < NT2; t += NW) {
        f32x4 wn[NT1];
#pragma unroll
        for (int s = 0; s < NT1; s++) wn[s] = load_w4<VEC>(W2, (t + NW) * 16 + nn, t + NW < NT2 ? h2 : 0, 16 * s + 4 * q, h1);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NT1; s++) {
            if (s & 1) acc1 = mfma4(w2[s], H1[s * 4 + q][nn], acc1);
            else acc0 = mfma4(w2[s], H1[s * 4 + q][nn], acc0);
        }
        const f32x4 hq = bias_relu(acc0 + acc1, b2, t * 16 + 4 * q, h2);
        H2[t * 4 + q][nn] = hq;
        if (h2_out && row_ok && t * 16 + 4 * q < h2) *(f32x4*)(h2_out + (long)row * h2 + t * 16 + 4 * q) = hq;
#pragma unroll
        for (int s = 0; s < NT1; s++) w2[s] = wn[s];
    }
    __syncthreads();

    // layer 3: out_dim <= 4 outputs = rows 0..3 of ONE tile (quarter q = 0); the waves split the k-steps
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < (NT2 + NW - 1) / NW; j++)
            if (wave + NW * j < NT2) acc = mfma4(w3[j], H2[(wave + NW * j) * 4 + q][nn], acc);
        if (q == 0) P[wave][nn] = acc;
    }
    __syncthreads();
    if (wave == 0 && q == 0 && row_ok) {
        z4 = P[0][nn];
#pragma unroll
        for (int w = 1; w < NW; w++) z4 += P[w][nn];
        return true;
    }
    return false;
}

// The same three layers by ONE wave for the first NR rows of a tile (the free-running rollout kernel's waves each own four envs and
// never meet a barrier): every output element is the chain of MFMAs mlp3_rows16 runs for it - same k order, the same two accumulators
// per tile, layer 3 as NW partial sums added in wave order - so a row's result is the 4-wave kernel's bit for bit (an MFMA's output
// column depends on its own B column only; columns >= NR read zeros and are discarded).  H1 / H2: the wave's own scratch,
// [NT1 * 4][NR], [NT2 * 4][NR] float4.  Returns true on the lanes that hold a row's layer-3 sums (first quarter, row valid).
template <int NT1, int NT2, bool VEC, int NR>
__device__ __forceinline__ bool mlp3_rows_wave(const int lane, const long row, int in_dim, int h1, int h2, int out_dim, const float* __restrict__ xa, int lda,
                                               const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                                               const float* __restrict__ b2, const float* __restrict__ W3, f32x4 (*H1)[NR], f32x4 (*H2)[NR], f32x4& z4) {
    const int nn = lane & 15, q = lane >> 4;
    const bool row_ok = row >= 0, col_ok = nn < NR;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 bx[KS_IN_MAX];
#pragma unroll
    for (int s = 0; s < KS_IN_MAX; s++) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = 16 * s + 4 * q + j;
            v[j] = (row_ok && k < in_dim) ? xa[(long)row * lda + k] : 0.f;
        }
        bx[s] = f32x4{v[0], v[1], v[2], v[3]};
    }
    f32x4 w2[NT1];
#pragma unroll
    for (int s = 0; s < NT1; s++) w2[s] = load_w4<VEC>(W2, nn, h2, 16 * s + 4 * q, h1);
    {
        f32x4 w1[KS_IN_MAX];
#pragma unroll
        for (int s = 0; s < KS_IN_MAX; s++) w1[s] = load_w4<false>(W1, nn, h1, 16 * s + 4 * q, in_dim);
        for (int t = 0; t < NT1; t++) {
            f32x4 wn[KS_IN_MAX];
#pragma unroll
            for (int s = 0; s < KS_IN_MAX; s++) wn[s] = load_w4<false>(W1, (t + 1) * 16 + nn, t + 1 < NT1 ? h1 : 0, 16 * s + 4 * q, in_dim);
            f32x4 acc0 = zero, acc1 = zero;
#pragma unroll
            for (int s = 0; s < KS_IN_MAX; s++) {
                if (s & 1) acc1 = mfma4(w1[s], bx[s], acc1);
                else acc0 = mfma4(w1[s], bx[s], acc0);
            }
            const f32x4 hq = bias_relu(acc0 + acc1, b1, t * 16 + 4 * q, h1);
            if (col_ok) H1[t * 4 + q][nn] = hq;
#pragma unroll
            for (int s = 0; s < KS_IN_MAX; s++) w1[s] = wn[s];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // layer 2: the B operands are the same for every tile - read once
    f32x4 hb[NT1];
#pragma unroll
    for (int s = 0; s < NT1; s++) hb[s] = col_ok ? H1[s * 4 + q][nn] : zero;
    for (int t = 0; t < NT2; t++) {
        f32x4 wn[NT1];
#pragma unroll
        for (int s = 0; s < NT1; s++) wn[s] = load_w4<VEC>(W2, (t + 1) * 16 + nn, t + 1 < NT2 ? h2 : 0, 16 * s + 4 * q, h1);
        f32x4 acc0 = zero, acc1 = zero;
#pragma unroll
        for (int s = 0; s < NT1; s++) {
            if (s & 1) acc1 = mfma4(w2[s], hb[s], acc1);
            else acc0 = mfma4(w2[s], hb[s], acc0);
        }
        const f32x4 hq = bias_relu(acc0 + acc1, b2, t * 16 + 4 * q, h2);
        if (col_ok) H2[t * 4 + q][nn] = hq;
#pragma unroll
        for (int s = 0; s < NT1; s++) w2[s] = wn[s];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // layer 3: the NW partial sums of the 4-wave kernel (tiles w, w + NW, ...), added in wave order
    f32x4 part[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) {
        f32x4 acc = zero;
#pragma unroll
        for (int j = 0; j < (NT2 + NW - 1) / NW; j++)
            if (w + NW * j < NT2) acc = mfma4(load_w4<VEC>(W3, nn, out_dim, 16 * (w + NW * j) + 4 * q, h2), col_ok ? H2[(w + NW * j) * 4 + q][nn] : zero, acc);
        part[w] = acc;
    }
    if (q == 0 && row_ok && col_ok) {
        z4 = part[0];
#pragma unroll
        for (int w = 1; w < NW; w++) z4 += part[w];
        return true;
    }
    return false;
}

}  // namespace kmlp
