// ks_rollout.hip -- batched rollout / replay bookkeeping kernels (C ABI: include/kinova_rollout.h).
//
// Every kernel is HBM-streaming elementwise work on struct-of-rows tensors owned by PyTorch: one wavefront per
// env (or per sampled window row), lanes across the 82 observation columns, so that row reads/writes are
// coalesced 256-byte bursts.  No LDS, no atomics; the only cross-env step (FIFO ranks of the kept episodes) is a
// single-block scan.  The torch implementations in rollout.py / replay.py are the checkers of these kernels.
#include <hip/hip_runtime.h>

#include "../../include/kinova_rollout.h"
#include "../../include/kinova_sim.h"
#include "ks_select.h"

namespace {

constexpr int WAVE = 64;
constexpr int S = KR_STATE_DIM, A = KR_ACTION_DIM;

__global__ __launch_bounds__(256) void k_select_action(int n, const float* __restrict__ obs, const float* __restrict__ prev_obs,
                                                       const uint8_t* __restrict__ has_prev, const int64_t* __restrict__ t, uint8_t* ready,
                                                       const float* __restrict__ actor_out, const float* __restrict__ noise, float sigma,
                                                       float max_action, int skip_steps, float* __restrict__ action,
                                                       float* __restrict__ action_t, uint8_t* __restrict__ lifting) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float pi[A], nz[A];
#pragma unroll
    for (int k = 0; k < A; k++) { pi[k] = actor_out[(long)i * A + k]; nz[k] = noise[(long)i * A + k]; }
    krsel::select_one(i, n, pi, nz, obs, prev_obs, has_prev, t, ready, sigma, max_action, skip_steps, action, action_t, lifting);
}

// one wave per env
__global__ __launch_bounds__(WAVE) void k_store_transition(int n, int H, int n_steps, int auto_reset, int with_replay,
                                                           const float* __restrict__ sim_obs, const float* __restrict__ sim_final,
                                                           const float* __restrict__ sim_reward, const uint8_t* __restrict__ sim_done,
                                                           float* obs, float* prev_obs, uint8_t* has_prev, int64_t* t, uint8_t* ready,
                                                           const uint8_t* __restrict__ lifting, const float* __restrict__ action,
                                                           float* cur_state, float* cur_next, float* cur_action, float* cur_reward,
                                                           float* cur_not_done, int64_t* cur_len, float* reward_out, uint8_t* done_out,
                                                           uint8_t* keep) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const bool done = sim_done[i] != 0, lift = lifting[i] != 0;
    const float rew = sim_reward[i];
    const bool store = with_replay && !lift;
    const long len0 = with_replay ? cur_len[i] : 0;
    const long tt = len0 < H - 1 ? len0 : H - 1;
    const long row = ((long)i * H + tt);
    for (int c = lane; c < S; c += WAVE) {
        const float so = sim_obs[(long)i * S + c];
        const float st = obs[(long)i * S + c];
        const float nx = (done && auto_reset) ? sim_final[(long)i * S + c] : so;
        if (store) { cur_state[row * S + c] = st; cur_next[row * S + c] = nx; }
        prev_obs[(long)i * S + c] = done ? so : st;
        obs[(long)i * S + c] = so;
    }
    if (store && lane < A) cur_action[row * A + lane] = action[(long)i * A + lane];
    if (lane == 0) {
        long len1 = len0;
        if (store) {
            cur_reward[row] = rew;
            cur_not_done[row] = done ? 0.0f : 1.0f;
            len1 = tt + 1;
        }
        if (with_replay) {
            // the episode ended during the scripted lift: the last stored transition carries the outcome
            if (done && lift && len1 > 0) {
                cur_reward[(long)i * H + len1 - 1] = rew;
                cur_not_done[(long)i * H + len1 - 1] = 0.0f;
            }
            cur_len[i] = len1;
            keep[i] = done && (len1 - n_steps > 1);
        }
        has_prev[i] = !done;
        t[i] = done ? 0 : t[i] + 1;
        ready[i] = (ready[i] != 0) && !done;
        reward_out[i] = rew;
        done_out[i] = done;
    }
}

// One wavefront, no LDS (so that the launch can run while the stepping kernel holds the CUs' LDS): every lane counts a
// contiguous slice of the flags, the lane totals are scanned with shuffles, the ranks written back per slice.
__global__ __launch_bounds__(WAVE) void k_rank_episodes(int n, const uint8_t* __restrict__ keep, int64_t* __restrict__ rank, int64_t* total) {
    const int lane = threadIdx.x;
    const int per = (n + WAVE - 1) / WAVE, i0 = min(lane * per, n), i1 = min(i0 + per, n);
    int local = 0;
    for (int i = i0; i < i1; i++) local += keep[i] != 0;
    int incl = local;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    int before = incl - local;
    for (int i = i0; i < i1; i++) {
        before += keep[i] != 0;
        rank[i] = before;
    }
    if (lane == WAVE - 1) total[0] = incl;
}

// one wave per env; only kept envs move data
__global__ __launch_bounds__(WAVE) void k_commit_episodes(int n, int H, int capacity, const uint8_t* __restrict__ keep,
                                                          const int64_t* __restrict__ rank, const int64_t* __restrict__ head,
                                                          const float* __restrict__ cur_state, const float* __restrict__ cur_next,
                                                          const float* __restrict__ cur_action, const float* __restrict__ cur_reward,
                                                          const float* __restrict__ cur_not_done, const int64_t* __restrict__ cur_len,
                                                          float* ep_state, float* ep_next, float* ep_action, float* ep_reward,
                                                          float* ep_not_done, int64_t* ep_len) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n || keep[i] == 0) return;
    const long slot = (head[0] + rank[i] - 1) % capacity;
    const long src = (long)i * H, dst = slot * H;
    // 16-byte copies where the episode blocks are 16-byte aligned (H * 82 floats: even H), several loads in flight: at an
    // episode boundary all 4096 envs commit (84 MB) while the rollout waits - float by float this took 0.4 ms
    if ((H * S) % 4 == 0) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f* s0 = (const v4f*)(cur_state + src * S); const v4f* s1 = (const v4f*)(cur_next + src * S);
        v4f* d0 = (v4f*)(ep_state + dst * S); v4f* d1 = (v4f*)(ep_next + dst * S);
        const int n4 = H * S / 4;
        int k = lane;
        for (; k + WAVE < n4; k += 2 * WAVE) {
            const v4f a = s0[k], b = s0[k + WAVE], c = s1[k], d = s1[k + WAVE];
            d0[k] = a; d0[k + WAVE] = b; d1[k] = c; d1[k + WAVE] = d;
        }
        for (; k < n4; k += WAVE) { d0[k] = s0[k]; d1[k] = s1[k]; }
    } else {
        for (int k = lane; k < H * S; k += WAVE) {
            ep_state[dst * S + k] = cur_state[src * S + k];
            ep_next[dst * S + k] = cur_next[src * S + k];
        }
    }
    for (int k = lane; k < H * A; k += WAVE) ep_action[dst * A + k] = cur_action[src * A + k];
    for (int k = lane; k < H; k += WAVE) {
        ep_reward[dst + k] = cur_reward[src + k];
        ep_not_done[dst + k] = cur_not_done[src + k];
    }
    if (lane == 0) ep_len[slot] = cur_len[i];
}

__global__ __launch_bounds__(256) void k_advance_ring(int n, int capacity, const int64_t* __restrict__ total, int64_t* head, int64_t* count,
                                                      const uint8_t* __restrict__ ended, int64_t* cur_len) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        const long k = total[0];
        head[0] = (head[0] + k) % capacity;
        const long c = count[0] + k;
        count[0] = c < capacity ? c : capacity;
    }
    if (i < n && ended[i] != 0) cur_len[i] = 0;
}

struct Ring {                      // one episode ring of DeviceEpisodeReplay (kr_ring of the C ABI)
    const int64_t* count;
    const int64_t* head;
    int capacity;
    const int64_t* ep_len;
    const float *ep_state, *ep_next, *ep_action, *ep_reward, *ep_not_done;
};

// one wave per window row (b, w); episodes b < B_agent are drawn from ring `ra`, the others from ring `re` (the expert
// demonstrations DDPGfD mixes into every batch, DDPGfD.py:232-254; B_agent = B: one ring)
__global__ __launch_bounds__(WAVE) void k_sample_windows(int B, int B_agent, int H, int n_steps, Ring ra, Ring re, const float* __restrict__ u_ep,
                                                         const float* __restrict__ u_start, unsigned long long seed,
                                                         const int64_t* __restrict__ draw, float* __restrict__ next_ends,
                                                         float* __restrict__ state, float* __restrict__ action, float* __restrict__ next_state,
                                                         float* __restrict__ reward, float* __restrict__ not_done, float* __restrict__ weight) {
    const int W = H - n_steps;
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= B * W) return;
    const int b = r / W, w = r % W;
    const Ring& g = b < B_agent ? ra : re;
    const int64_t *count = g.count, *head = g.head, *ep_len = g.ep_len;
    const int capacity = g.capacity;
    const float *ep_state = g.ep_state, *ep_next = g.ep_next, *ep_action = g.ep_action, *ep_reward = g.ep_reward, *ep_not_done = g.ep_not_done;
    // np.random.randint(replay_ep_num - 1): the k-th OLDEST episode, k in [0, count - 1) - the newest one is never sampled
    // (utils.py:259).  In the ring the oldest episode sits at head - count, so after the first wrap the excluded slot is
    // head - 1, wherever that is.  With fewer than two episodes there is nothing to sample: every row gets weight 0.
    const long cnt = count[0];
    const bool none = cnt < 2;
    long hi = cnt - 1;
    hi = hi > 1 ? hi : 1;
    // the uniforms: the caller's, or (u_ep == nullptr) Philox4x32-10 keyed by the seed at counter (draw[0], b | row, tag) -
    // `draw` is a device counter that does not change while this kernel runs (the learner's update count)
    float ue, us;
    if (u_ep != nullptr) { ue = u_ep[b]; us = u_start[(long)b * W + w]; }
    else {
        const unsigned long long d = (unsigned long long)draw[0];
        uint32_t r4[4];
        krsel::philox4x32((uint32_t)b, (uint32_t)d, (uint32_t)(d >> 32), 0x5a4du, (uint32_t)seed, (uint32_t)(seed >> 32), r4);
        ue = (float)(r4[0] >> 8) * (1.0f / 16777216.0f);
        krsel::philox4x32((uint32_t)r, (uint32_t)d, (uint32_t)(d >> 32), 0x5a4eu, (uint32_t)seed, (uint32_t)(seed >> 32), r4);
        us = (float)(r4[0] >> 8) * (1.0f / 16777216.0f);
    }
    long k = (long)(ue * (float)hi);
    k = k < hi - 1 ? k : hi - 1;
    long ep = (head[0] - cnt + k) % capacity;
    ep = ep < 0 ? ep + capacity : ep;
    long ceiling = ep_len[ep] - n_steps;
    ceiling = ceiling > 1 ? ceiling : 1;
    long start = (long)(us * (float)ceiling);
    start = start < H - n_steps ? start : H - n_steps;
    if (w == ceiling - 1) start = ceiling;     // the final window of the episode (utils.py:283-301)
    start = start < H - n_steps ? start : H - n_steps;
    const long src = ep * H + start, dst = (long)r * n_steps;
    for (int k = lane; k < n_steps * S; k += WAVE) {
        state[dst * S + k] = ep_state[src * S + k];
        next_state[dst * S + k] = ep_next[src * S + k];
    }
    // the rows the target networks evaluate (DDPGfD.py:256-275: next_state[:, 0] and next_state[:, -1]) as one [2 B W, 82] block
    if (next_ends != nullptr) {
        for (int k = lane; k < S; k += WAVE) {
            next_ends[(long)r * S + k] = ep_next[src * S + k];
            next_ends[((long)B * W + r) * S + k] = ep_next[(src + n_steps - 1) * S + k];
        }
    }
    for (int k = lane; k < n_steps * A; k += WAVE) action[dst * A + k] = ep_action[src * A + k];
    if (lane < n_steps) {
        reward[dst + lane] = ep_reward[src + lane];
        not_done[dst + lane] = ep_not_done[src + lane];
    }
    if (lane == 0) weight[r] = (!none && w < ceiling) ? 1.0f : 0.0f;
}

// ---- learner glue: plain grid-stride elementwise kernels, no fma contraction where the torch expression has none
// (ONE wave, no LDS: the update's body must be able to start beside a stepping kernel that holds ALL of a CU's LDS - with the larger hull
// tables of a mixed-object context not even the 1 KB of a block reduction is left, and a learner whose first kernel waits for LDS only
// runs when persistent workgroups exit: the episodes published meanwhile were dropped, round 4.  The sums are wave butterflies.)
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

__global__ __launch_bounds__(WAVE) void k_critic_grad(int R, int n, const float* __restrict__ q, const float* __restrict__ tq1,
                                                      const float* __restrict__ tqn, const float* __restrict__ reward,
                                                      const float* __restrict__ weight, const float* __restrict__ wsum, float discount,
                                                      float* __restrict__ dq, float* losses) {
#pragma clang fp contract(off)
    // single wave: the batch is a few thousand rows; the masked means are wave reductions
    const float inv = wsum[0] > 0.0f ? 1.0f / wsum[0] : 0.0f;     // an all-padding batch (empty replay) has zero loss and gradient
    float l1 = 0, ln = 0;
    for (int r = threadIdx.x; r < R; r += WAVE) {
        const float t1 = reward[(long)r * n] + discount * tq1[r];
        float ret = 0, g = 1.0f;
        for (int i = 0; i < n; i++) { ret += g * reward[(long)r * n + i]; g *= discount; }
        const float tn = ret + g * tqn[r];
        const float w = weight ? weight[r] : 1.0f, e1 = q[r] - t1, en = q[r] - tn;
        l1 += w * e1 * e1;
        ln += w * en * en;
        dq[r] = w * inv * (2.0f * e1 + 0.5f * 2.0f * en);
    }
    l1 = wave_sum(l1);
    ln = wave_sum(ln);
    if (threadIdx.x == 0) { losses[1] = l1 * inv; losses[2] = ln * inv; losses[0] = l1 * inv + 0.5f * (ln * inv); }
}

// start of an update's body: what used to be eight tiny library launches (sum, clamp, mul, div, copy, add, fill, copy)
__global__ __launch_bounds__(WAVE) void k_update_prologue(int R, int n, const float* __restrict__ weight, float* __restrict__ wsum,
                                                          float* __restrict__ dq_actor, int64_t* it, int64_t* it_head, int pipelined) {
#pragma clang fp contract(off)
    float s = 0;
    for (int r = threadIdx.x; r < R; r += WAVE) s += weight ? weight[r] : 1.0f;
    s = wave_sum(s);                                  // (0 / 1 weights: exact in any order; every lane holds the total)
    const float total = s > 1.0f ? s : 1.0f;          // exact unless the batch is all padding
    if (threadIdx.x == 0) {
        wsum[0] = total;
        it[0] += 1;                                   // this update's number (Adam bias correction, soft-update phase)
        if (pipelined) it_head[0] = it[0];            // its actor step is applied by the NEXT update's head
    }
    const float scale = -1.0f / (total * (float)n);   // d(-sum_r w_r sum_k Q_rk / (sum(w) n)) / dQ_rk
    for (int i = threadIdx.x; i < R * n; i += WAVE) dq_actor[i] = (weight ? weight[i / n] : 1.0f) * scale;
}

__global__ __launch_bounds__(256) void k_relu_backward(long count, const float* __restrict__ act, float* __restrict__ grad) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) grad[i] = act[i] > 0.0f ? grad[i] : 0.0f;
}

__global__ __launch_bounds__(256) void k_sigmoid_scale_backward(long count, const float* __restrict__ a, float max_action, float* __restrict__ grad) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) grad[i] *= a[i] * (1.0f - a[i] / max_action);
}

__global__ __launch_bounds__(256) void k_adam_step(long count, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, const int64_t* __restrict__ step, float lr, float b1, float b2,
                                                   float eps, float wd) {
#pragma clang fp contract(off)
    // torch.optim.Adam, single-tensor formulation: bias corrections from the (already incremented) device step
    if (step[0] <= 0) return;                 // no update has produced gradients yet
    const float t = (float)step[0];
    const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
    const float step_size = lr / bc1, bc2s = sqrtf(bc2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) {
        float gi = g[i];
        if (wd != 0.0f) gi = gi + wd * p[i];
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);            // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2s + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

__global__ __launch_bounds__(256) void k_soft_update(long count, const float* __restrict__ p, float* __restrict__ tp, float tau,
                                                     const int64_t* __restrict__ it, int freq) {
#pragma clang fp contract(off)
    if (it[0] <= 0 || it[0] % freq != 0) return;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) tp[i] = tau * p[i] + (1.0f - tau) * tp[i];
}

// one wave, no LDS: every lane scans its share of the values (agent-scope loads: the writers' stores are device-visible), the wave takes
// the minimum with shuffles; the launch ends when it has reached the target or the wall clock runs out
__global__ __launch_bounds__(64) void k_wait_min(const int64_t* __restrict__ values, int n, int64_t target, long long ticks, int64_t* timeouts) {
    const long long t0 = wall_clock64();
    for (;;) {
        long long m = 0x7fffffffffffffffll;
        for (int i = threadIdx.x; i < n; i += 64) {
            const long long v = __hip_atomic_load(values + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            m = v < m ? v : m;
        }
        for (int o = 32; o > 0; o >>= 1) {
            const long long other = __shfl_xor(m, o);
            m = other < m ? other : m;
        }
        if (m >= target) break;                                                    // (wave-uniform)
        if (ticks > 0 && wall_clock64() - t0 > ticks) {
            if (timeouts && threadIdx.x == 0) atomicAdd((unsigned long long*)timeouts, 1ull);   // a time-out is never silent
            break;
        }
        __builtin_amdgcn_s_sleep(64);
    }
}

inline int grid_for(long count) { long b = (count + 255) / 256; return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); }

inline int launched() { return hipGetLastError() == hipSuccess ? KS_OK : KS_ERR_HIP; }

}  // namespace

extern "C" {

int kr_select_action(int32_t n, const float* obs, const float* prev_obs, const uint8_t* has_prev, const int64_t* t, uint8_t* ready,
                     const float* actor_out, const float* noise, float sigma, float max_action, int32_t skip_steps, float* action,
                     float* action_t, uint8_t* lifting, void* stream) {
    if (n <= 0 || !obs || !prev_obs || !has_prev || !t || !ready || !actor_out || !noise || !action || !action_t || !lifting) return KS_ERR_INVALID;
    hipLaunchKernelGGL(k_select_action, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, obs, prev_obs, has_prev, t, ready, actor_out,
                       noise, sigma, max_action, skip_steps, action, action_t, lifting);
    return launched();
}

int kr_store_transition(int32_t n, int32_t horizon, int32_t n_steps, int32_t auto_reset, int32_t with_replay, const float* sim_obs,
                        const float* sim_final_obs, const float* sim_reward, const uint8_t* sim_done, float* obs, float* prev_obs,
                        uint8_t* has_prev, int64_t* t, uint8_t* ready, const uint8_t* lifting, const float* action, float* cur_state,
                        float* cur_next, float* cur_action, float* cur_reward, float* cur_not_done, int64_t* cur_len, float* reward_out,
                        uint8_t* done_out, uint8_t* keep, void* stream) {
    if (n <= 0 || !sim_obs || !sim_reward || !sim_done || !obs || !prev_obs || !has_prev || !t || !ready || !lifting || !reward_out || !done_out)
        return KS_ERR_INVALID;
    if (auto_reset && !sim_final_obs) return KS_ERR_INVALID;
    if (with_replay && (!action || !cur_state || !cur_next || !cur_action || !cur_reward || !cur_not_done || !cur_len || !keep)) return KS_ERR_INVALID;
    hipLaunchKernelGGL(k_store_transition, dim3(n), dim3(WAVE), 0, (hipStream_t)stream, n, horizon, n_steps, auto_reset, with_replay, sim_obs,
                       sim_final_obs, sim_reward, sim_done, obs, prev_obs, has_prev, t, ready, lifting, action, cur_state, cur_next, cur_action,
                       cur_reward, cur_not_done, cur_len, reward_out, done_out, keep);
    return launched();
}

int kr_wait_min_counted(const int64_t* values, int32_t n, int64_t target, double timeout_s, int64_t* timeouts, void* stream) {
    if (!values || n <= 0) return KS_ERR_INVALID;
    hipLaunchKernelGGL(k_wait_min, dim3(1), dim3(64), 0, (hipStream_t)stream, values, n, target, (long long)(timeout_s * 1e8), timeouts);   // wall_clock64: 100 MHz
    return launched();
}

int kr_wait_min(const int64_t* values, int32_t n, int64_t target, double timeout_s, void* stream) {
    return kr_wait_min_counted(values, n, target, timeout_s, nullptr, stream);
}

int kr_rank_episodes(int32_t n, const uint8_t* keep, int64_t* rank, int64_t* total, void* stream) {
    if (n <= 0 || !keep || !rank || !total) return KS_ERR_INVALID;
    hipLaunchKernelGGL(k_rank_episodes, dim3(1), dim3(WAVE), 0, (hipStream_t)stream, n, keep, rank, total);
    return launched();
}

int kr_commit_episodes(int32_t n, int32_t horizon, int32_t capacity, const uint8_t* keep, const int64_t* rank, const int64_t* head,
                       const float* cur_state, const float* cur_next, const float* cur_action, const float* cur_reward,
                       const float* cur_not_done, const int64_t* cur_len, float* ep_state, float* ep_next, float* ep_action, float* ep_reward,
                       float* ep_not_done, int64_t* ep_len, void* stream) {
    if (n <= 0 || capacity <= 0 || !keep || !rank || !head || !cur_state || !cur_next || !cur_action || !cur_reward || !cur_not_done || !cur_len ||
        !ep_state || !ep_next || !ep_action || !ep_reward || !ep_not_done || !ep_len)
        return KS_ERR_INVALID;
    hipLaunchKernelGGL(k_commit_episodes, dim3(n), dim3(WAVE), 0, (hipStream_t)stream, n, horizon, capacity, keep, rank, head, cur_state, cur_next,
                       cur_action, cur_reward, cur_not_done, cur_len, ep_state, ep_next, ep_action, ep_reward, ep_not_done, ep_len);
    return launched();
}

int kr_advance_ring(int32_t n, int32_t capacity, const int64_t* total, int64_t* head, int64_t* count, const uint8_t* ended, int64_t* cur_len,
                    void* stream) {
    if (n <= 0 || capacity <= 0 || !total || !head || !count || !ended || !cur_len) return KS_ERR_INVALID;
    hipLaunchKernelGGL(k_advance_ring, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, capacity, total, head, count, ended, cur_len);
    return launched();
}

int kr_sample_windows(int32_t batch, int32_t horizon, int32_t n_steps, const int64_t* count, const int64_t* head, int32_t capacity,
                      const int64_t* ep_len, const float* u_ep, const float* u_start, const float* ep_state, const float* ep_next, const float* ep_action, const float* ep_reward,
                      const float* ep_not_done, float* state, float* action, float* next_state, float* reward, float* not_done, float* weight,
                      void* stream) {
    if (batch <= 0 || horizon <= n_steps || n_steps <= 0 || n_steps > WAVE || capacity <= 0 || !count || !head || !ep_len || !u_ep || !u_start || !ep_state || !ep_next ||
        !ep_action || !ep_reward || !ep_not_done || !state || !action || !next_state || !reward || !not_done || !weight)
        return KS_ERR_INVALID;
    const Ring g{count, head, capacity, ep_len, ep_state, ep_next, ep_action, ep_reward, ep_not_done};
    hipLaunchKernelGGL(k_sample_windows, dim3(batch * (horizon - n_steps)), dim3(WAVE), 0, (hipStream_t)stream, batch, batch, horizon, n_steps, g, g,
                       u_ep, u_start, 0ull, (const int64_t*)nullptr, (float*)nullptr, state, action, next_state, reward, not_done, weight);
    return launched();
}

int kr_sample_windows_draw(int32_t batch, int32_t horizon, int32_t n_steps, const int64_t* count, const int64_t* head, int32_t capacity,
                           const int64_t* ep_len, uint64_t seed, const int64_t* draw, const float* ep_state, const float* ep_next,
                           const float* ep_action, const float* ep_reward, const float* ep_not_done, float* state, float* action, float* next_state,
                           float* reward, float* not_done, float* weight, float* next_ends, void* stream) {
    if (batch <= 0 || horizon <= n_steps || n_steps <= 0 || n_steps > WAVE || capacity <= 0 || !count || !head || !ep_len || !draw || !ep_state ||
        !ep_next || !ep_action || !ep_reward || !ep_not_done || !state || !action || !next_state || !reward || !not_done || !weight)
        return KS_ERR_INVALID;
    const Ring g{count, head, capacity, ep_len, ep_state, ep_next, ep_action, ep_reward, ep_not_done};
    hipLaunchKernelGGL(k_sample_windows, dim3(batch * (horizon - n_steps)), dim3(WAVE), 0, (hipStream_t)stream, batch, batch, horizon, n_steps, g, g,
                       (const float*)nullptr, (const float*)nullptr, (unsigned long long)seed, draw, next_ends, state, action, next_state, reward,
                       not_done, weight);
    return launched();
}

static bool ring_ok(const kr_ring* r) {
    return r && r->count && r->head && r->capacity > 0 && r->ep_len && r->ep_state && r->ep_next && r->ep_action && r->ep_reward && r->ep_not_done;
}

int kr_sample_windows_mixed(int32_t batch, int32_t batch_agent, int32_t horizon, int32_t n_steps, const kr_ring* agent, const kr_ring* expert,
                            const float* u_ep, const float* u_start, uint64_t seed, const int64_t* draw, float* state, float* action,
                            float* next_state, float* reward, float* not_done, float* weight, float* next_ends, void* stream) {
    if (batch <= 0 || batch_agent < 0 || batch_agent > batch || horizon <= n_steps || n_steps <= 0 || n_steps > WAVE || !ring_ok(agent) || !ring_ok(expert) ||
        ((u_ep == nullptr) != (u_start == nullptr)) || (!u_ep && !draw) || !state || !action || !next_state || !reward || !not_done || !weight)
        return KS_ERR_INVALID;
    const Ring ga{agent->count, agent->head, agent->capacity, agent->ep_len, agent->ep_state, agent->ep_next, agent->ep_action, agent->ep_reward,
                  agent->ep_not_done};
    const Ring ge{expert->count, expert->head, expert->capacity, expert->ep_len, expert->ep_state, expert->ep_next, expert->ep_action,
                  expert->ep_reward, expert->ep_not_done};
    hipLaunchKernelGGL(k_sample_windows, dim3(batch * (horizon - n_steps)), dim3(WAVE), 0, (hipStream_t)stream, batch, batch_agent, horizon, n_steps, ga,
                       ge, u_ep, u_start, (unsigned long long)seed, draw, next_ends, state, action, next_state, reward, not_done, weight);
    return launched();
}

int kr_critic_grad(int32_t rows, int32_t n_steps, const float* q, const float* tq1, const float* tqn, const float* reward, const float* weight,
                   const float* weight_sum, float discount, float* dq, float* losses, void* stream) {
    if (rows <= 0 || n_steps <= 0 || !q || !tq1 || !tqn || !reward || !weight_sum || !dq || !losses) return KS_ERR_INVALID;
    hipLaunchKernelGGL(k_critic_grad, dim3(1), dim3(WAVE), 0, (hipStream_t)stream, rows, n_steps, q, tq1, tqn, reward, weight, weight_sum, discount, dq,
                       losses);
    return launched();
}

int kr_update_prologue(int32_t rows, int32_t n_steps, const float* weight, float* weight_sum, float* dq_actor, int64_t* it, int64_t* it_head,
                       int32_t pipelined, void* stream) {
    if (rows <= 0 || n_steps <= 0 || !weight_sum || !dq_actor || !it || !it_head) return KS_ERR_INVALID;
    hipLaunchKernelGGL(k_update_prologue, dim3(1), dim3(WAVE), 0, (hipStream_t)stream, rows, n_steps, weight, weight_sum, dq_actor, it, it_head, pipelined);
    return launched();
}

int kr_relu_backward(int64_t count, const float* act, float* grad, void* stream) {
    if (count <= 0 || !act || !grad) return KS_ERR_INVALID;
    hipLaunchKernelGGL(k_relu_backward, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, (long)count, act, grad);
    return launched();
}

int kr_sigmoid_scale_backward(int64_t count, const float* a, float max_action, float* grad, void* stream) {
    if (count <= 0 || !a || !grad) return KS_ERR_INVALID;
    hipLaunchKernelGGL(k_sigmoid_scale_backward, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, (long)count, a, max_action, grad);
    return launched();
}

int kr_adam_step(int64_t count, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const int64_t* step, float lr, float beta1,
                 float beta2, float eps, float weight_decay, void* stream) {
    if (count <= 0 || !param || !grad || !exp_avg || !exp_avg_sq || !step) return KS_ERR_INVALID;
    hipLaunchKernelGGL(k_adam_step, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, (long)count, param, grad, exp_avg, exp_avg_sq, step, lr,
                       beta1, beta2, eps, weight_decay);
    return launched();
}

int kr_soft_update(int64_t count, const float* param, float* target, float tau, const int64_t* it, int32_t freq, void* stream) {
    if (count <= 0 || freq <= 0 || !param || !target || !it) return KS_ERR_INVALID;
    hipLaunchKernelGGL(k_soft_update, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, (long)count, param, target, tau, it, freq);
    return launched();
}

}  // extern "C"
