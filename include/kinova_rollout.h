/* include/kinova_rollout.h -- C ABI of the batched rollout / replay kernels in libkinova_sim.so.
 *
 * The reference drives ONE env from Python and keeps its replay in Python lists; for N envs in lock step the same
 * bookkeeping is a handful of per-env elementwise rules.  Each entry point below is one kernel launch that replaces
 * the loop body named beside it (reference root /root/reference/gym-kinova-gripper):
 *
 *   kr_select_action    main_DDPGfD.py:425-451  check_grasp latch (expert_data.py:559-593), exploration noise,
 *                                               clip, scripted lift action
 *   kr_store_transition main_DDPGfD.py:443-471  replay_buffer.add (utils.py:34-64) for envs that are not lifting,
 *                                               replace (utils.py:309-343) when an episode ends during the lift,
 *                                               per-episode counters; decides which episodes are kept (len-n > 1)
 *   kr_rank_episodes    utils.py:66-90          FIFO slot of every kept episode (env order)
 *   kr_commit_episodes  utils.py:66-90          copy the kept open episodes into the episode ring
 *   kr_advance_ring     utils.py:66-90          ring head / count, open-episode lengths of finished envs
 *   kr_sample_windows   utils.py:240-306        sample_batch_nstep: per sampled episode, ceiling-1 uniform window
 *                                               starts + the final window, as ONE fixed-shape padded batch
 *
 * Conventions: all pointers are DEVICE pointers owned by the caller (PyTorch tensors); bool arrays are one byte per
 * element; counters are int64; float data is fp32, row-major; every call is asynchronous on `stream` (hipStream_t
 * as void*) and does no host synchronisation, so sequences of them can be captured in a HIP graph.
 * Return value: 0, or a negative ks_status (kinova_sim.h).
 */
#ifndef KINOVA_ROLLOUT_H
#define KINOVA_ROLLOUT_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KR_STATE_DIM 82
#define KR_ACTION_DIM 4

/* action selection for n envs.  obs/prev_obs [n,82], has_prev/ready/lifting bool [n], t int64 [n] (steps taken in
 * the episode), actor_out [n,4] = pi(obs), noise [n,4] ~ N(0,1).  ready is updated in place (latched),
 * action [n,4], action_t [4,n] (the layout ks_step takes) and lifting (= ready at the time of the action) are written. */
int kr_select_action(int32_t n, const float *obs, const float *prev_obs, const uint8_t *has_prev, const int64_t *t, uint8_t *ready,
                     const float *actor_out, const float *noise, float sigma, float max_action, int32_t skip_steps,
                     float *action, float *action_t, uint8_t *lifting, void *stream);

/* replay write + bookkeeping after ks_step.  sim_obs/sim_final_obs [n,82], sim_reward [n], sim_done uint8 [n]
 * (non-zero = finished), auto_reset as in ks_config.  Engine state (in/out): obs (the state the action was taken
 * in; becomes the new observation), prev_obs, has_prev, t, ready; lifting/action from kr_select_action.
 * Open episodes (in/out): cur_state/cur_next [n,H,82], cur_action [n,H,4], cur_reward/cur_not_done [n,H],
 * cur_len int64 [n].  Outputs: reward_out [n], done_out bool [n], keep bool [n] (finished AND len - n_steps > 1). */
int kr_store_transition(int32_t n, int32_t horizon, int32_t n_steps, int32_t auto_reset, int32_t with_replay,
                        const float *sim_obs, const float *sim_final_obs, const float *sim_reward, const uint8_t *sim_done,
                        float *obs, float *prev_obs, uint8_t *has_prev, int64_t *t, uint8_t *ready, const uint8_t *lifting,
                        const float *action, float *cur_state, float *cur_next, float *cur_action, float *cur_reward,
                        float *cur_not_done, int64_t *cur_len, float *reward_out, uint8_t *done_out, uint8_t *keep, void *stream);

/* rank [n] int64: number of kept episodes among envs 0..i (inclusive); total int64 [1] */
int kr_rank_episodes(int32_t n, const uint8_t *keep, int64_t *rank, int64_t *total, void *stream);
/* Stream-side pacing for the free-running rollout (no reference counterpart: main_DDPGfD.py:424-486 alternates acting and learning on one
 * thread): a one-wave, LDS-free kernel on `stream` that returns when min_i values[i] >= target (ks_rollout_args.steps_total: every env has
 * done that many env-steps of the persistent rollout launch) or after timeout_s of wall clock - whatever follows on the stream (the
 * learner's next update, the commit of published episodes) then stays within a bounded distance of the SLOWEST env however long the launch is.
 * timeout_s <= 0 waits for ever (like a library collective: only for values that are certain to arrive).  kr_wait_min_counted also adds 1
 * to *timeouts (device int64, may be NULL) when the wait ended on the clock instead of the target, so that a paced run that degrades - e.g.
 * more stepping workgroups than resident slots, whose second round only starts when the first has finished its whole launch - is visible. */
int kr_wait_min(const int64_t *values, int32_t n, int64_t target, double timeout_s, void *stream);
int kr_wait_min_counted(const int64_t *values, int32_t n, int64_t target, double timeout_s, int64_t *timeouts, void *stream);

/* ring rows ep_* [capacity(+), H, ...], ep_len int64; kept env i goes to slot (head + rank[i] - 1) % capacity */
int kr_commit_episodes(int32_t n, int32_t horizon, int32_t capacity, const uint8_t *keep, const int64_t *rank, const int64_t *head,
                       const float *cur_state, const float *cur_next, const float *cur_action, const float *cur_reward,
                       const float *cur_not_done, const int64_t *cur_len, float *ep_state, float *ep_next, float *ep_action,
                       float *ep_reward, float *ep_not_done, int64_t *ep_len, void *stream);

/* head = (head + total) % capacity, count = min(capacity, count + total), cur_len[i] = 0 where ended[i] */
int kr_advance_ring(int32_t n, int32_t capacity, const int64_t *total, int64_t *head, int64_t *count, const uint8_t *ended,
                    int64_t *cur_len, void *stream);

/* n-step window batch: B episodes x W = horizon - n_steps rows.  u_ep [B], u_start [B,W] uniform in [0,1).
 * Episode b is the floor(u_ep[b] * (count - 1))-th OLDEST of the ring (slot head - count + k mod capacity): the newest
 * episode is never sampled (np.random.randint(replay_ep_num - 1), utils.py:259), before and after the ring wraps.
 * Outputs state/next_state [B*W,n_steps,82], action [B*W,n_steps,4], reward/not_done [B*W,n_steps], weight [B*W]
 * (1 for real windows, 0 for padding rows; all 0 while the ring holds fewer than two episodes). */
int kr_sample_windows(int32_t batch, int32_t horizon, int32_t n_steps, const int64_t *count, const int64_t *head, int32_t capacity,
                      const int64_t *ep_len, const float *u_ep, const float *u_start, const float *ep_state, const float *ep_next, const float *ep_action, const float *ep_reward,
                      const float *ep_not_done, float *state, float *action, float *next_state, float *reward, float *not_done,
                      float *weight, void *stream);

/* The same batch with the uniforms drawn in the kernel: Philox4x32-10 keyed by `seed`, counter (draw[0], episode b | row) -
 * `draw` is a device counter that is constant while the kernel runs and differs from call to call (the learner's update
 * count).  No host-side generator, so a captured graph needs no generator-state launches.  next_ends (optional) [2 B W, 82]:
 * next_state[:, 0] followed by next_state[:, n_steps - 1], the rows the target networks evaluate (DDPGfD.py:256-275). */
int kr_sample_windows_draw(int32_t batch, int32_t horizon, int32_t n_steps, const int64_t *count, const int64_t *head, int32_t capacity,
                           const int64_t *ep_len, uint64_t seed, const int64_t *draw, const float *ep_state, const float *ep_next,
                           const float *ep_action, const float *ep_reward, const float *ep_not_done, float *state, float *action,
                           float *next_state, float *reward, float *not_done, float *weight, float *next_ends, void *stream);

/* DDPGfD's demonstration mix (DDPGfD.train_batch, DDPGfD.py:232-254: agent_batch_size = int(batch_size * (1 - prob)) episodes from
 * the agent's replay, the other batch_size - agent_batch_size from the expert replay, concatenated agent first) as ONE launch:
 * episodes b < batch_agent of the batch are drawn from ring `agent`, the others from ring `expert`, each with the rule of
 * kr_sample_windows on its own ring (k-th oldest of count - 1; a ring with fewer than two episodes yields weight-0 rows).
 * Both rings have row shape [*, horizon, ...].  Uniforms: u_ep [batch] + u_start [batch, W] when given (tests), else drawn in the
 * kernel exactly as kr_sample_windows_draw does (Philox keyed by seed, draw[0], b | row).  next_ends optional as there. */
typedef struct {
    const int64_t *count, *head;      /* device scalars: committed episodes, next slot */
    int32_t capacity;
    const int64_t *ep_len;            /* [capacity(+)] */
    const float *ep_state, *ep_next, *ep_action, *ep_reward, *ep_not_done;
} kr_ring;
int kr_sample_windows_mixed(int32_t batch, int32_t batch_agent, int32_t horizon, int32_t n_steps, const kr_ring *agent, const kr_ring *expert,
                            const float *u_ep, const float *u_start, uint64_t seed, const int64_t *draw, float *state, float *action,
                            float *next_state, float *reward, float *not_done, float *weight, float *next_ends, void *stream);

/* ---- learner glue (DDPGfD.train_batch, DDPGfD.py:219-367): the elementwise steps between the GEMMs, one launch each
 *
 *   kr_critic_grad   targets + dLoss/dQ of the critic loss L1 + 0.5 LN with masked row means (DDPGfD.py:256-330):
 *                      target_Q  = r[:,0] + discount * tq1,   target_QN = sum_i discount^i r[:,i] + discount^n * tqn
 *                      dq = w / sum(w) * (2 (q - target_Q) + 0.5 * 2 (q - target_QN));  losses[0..2] = (loss, L1, LN)
 *   kr_relu_backward g = (z > 0) ? g : 0 in place (z = the ReLU OUTPUT of the layer)
 *   kr_sigmoid_scale_backward  g *= a (1 - a / max_action)  for a = max_action * sigmoid(z)   (DDPGfD.py:32)
 *   kr_adam_step     torch.optim.Adam (lr, betas, eps, weight_decay as L2 added to the gradient) on a flat parameter buffer;
 *                      step is a device counter that the caller has already incremented
 *   kr_soft_update   target = tau * p + (1 - tau) * target on every `freq`-th value of the device counter `it`
 *                      (DDPGfD.py:360-366), else unchanged
 */
int kr_critic_grad(int32_t rows, int32_t n_steps, const float *q, const float *tq1, const float *tqn, const float *reward,
                   const float *weight, const float *weight_sum, float discount, float *dq, float *losses, void *stream);
/* start of an update (one launch): weight_sum[0] = max(sum(weight), 1) (weight NULL = all ones), dq_actor[rows * n_steps] =
 * -weight[row] / (weight_sum * n_steps) (dLoss/dQ of the actor loss -mean Q(s, pi(s)), DDPGfD.py:341), it[0] += 1 (the update
 * counter Adam and the soft update read), and with pipelined != 0 it_head[0] = it[0]: this update's actor step will be applied by
 * the head of the next one (learner_native.phase_head). */
int kr_update_prologue(int32_t rows, int32_t n_steps, const float *weight, float *weight_sum, float *dq_actor, int64_t *it, int64_t *it_head,
                       int32_t pipelined, void *stream);
int kr_relu_backward(int64_t count, const float *act, float *grad, void *stream);
int kr_sigmoid_scale_backward(int64_t count, const float *a, float max_action, float *grad, void *stream);
int kr_adam_step(int64_t count, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, const int64_t *step, float lr,
                 float beta1, float beta2, float eps, float weight_decay, void *stream);
int kr_soft_update(int64_t count, const float *param, float *target, float tau, const int64_t *it, int32_t freq, void *stream);

/* ---- fused 3-layer MLP forward on the matrix cores (fp32 MFMA): Actor.forward (DDPGfD.py:29-32) and
 * Critic.forward (DDPGfD.py:47-50) as ONE launch each:
 *     out[n,out_dim] = f(W3 relu(W2 relu(W1 x + b1) + b2) + b3),   f = identity (KR_ACT_NONE) or scale * sigmoid
 * x = the first in_a columns from xa ([n, >= in_a], row stride lda) followed by in_b columns from xb (row stride ldb;
 * in_b = 0: xb unused) - the critic's cat([state, action], 1) without materialising it.  Weights in torch.nn.Linear
 * layout (W [out][in] row-major, b [out]).  Limits: in_a + in_b <= 96, out_dim <= 4, hidden widths 256-256 (BASELINE),
 * 400-300 (reference), 128-128, 64-64; any other width returns KS_ERR_INVALID and the caller keeps its GEMM path.
 * h1_out [n,h1] / h2_out [n,h2] (optional, NULL to skip): the hidden activations (after the ReLU), which the backward
 * pass of a training step needs; they require h % 4 == 0 and 16-byte aligned buffers. */
#define KR_ACT_NONE 0
#define KR_ACT_SIGMOID 1
int kr_mlp3_forward(int32_t n, int32_t in_a, int32_t in_b, int32_t h1, int32_t h2, int32_t out_dim, const float *xa, int32_t lda,
                    const float *xb, int32_t ldb, const float *W1, const float *b1, const float *W2, const float *b2, const float *W3,
                    const float *b3, int32_t act, float scale, float *out, float *h1_out, float *h2_out, void *stream);

/* The same forward WITHOUT LDS (one wavefront per 16 rows, everything in registers, <= 168 registers per lane): its
 * waves can be resident beside a kernel that holds a CU's whole LDS (k_env_step), so a learner's forward-only passes
 * run in that kernel's shadow instead of behind it.  Widths 256-256, 128-128, 64-64; otherwise KS_ERR_INVALID. */
int kr_mlp3_forward_shadow(int32_t n, int32_t in_a, int32_t in_b, int32_t h1, int32_t h2, int32_t out_dim, const float *xa, int32_t lda,
                           const float *xb, int32_t ldb, const float *W1, const float *b1, const float *W2, const float *b2,
                           const float *W3, const float *b3, int32_t act, float scale, float *out, float *h1_out, float *h2_out,
                           void *stream);

/* Backward of the same MLP, LDS-free like kr_mlp3_forward_shadow (DDPGfD.train_batch's loss.backward(), DDPGfD.py:330-356):
 *   kr_mlp3_backward_shadow   data gradients  dz2 = (dz3 W3) * [h2 > 0],  dz1 = (dz2 W2) * [h1 > 0]  ([n,h2], [n,h1];
 *                             either may be NULL) and optionally  dx = dz1 W1[:, col0 : col0 + ncol]  ([n,ncol], ncol <= 4:
 *                             dQ/da through the critic's first layer), followed - when act_out [n,ncol] is given - by the
 *                             backward of a = scale * sigmoid(z):  dx *= a (1 - a / scale).  dz3 [n,out_dim], h1 / h2 the
 *                             activations kr_mlp3_forward* stored.  Widths: multiples of 16 up to 256.
 *   kr_weight_grad_shadow     dW [M,N] = dz^T [ha | hb]  and  db [M] = column sums of dz, dz [n,M], ha [n,>=Na] (row stride
 *                             lda), hb [n,>=Nb] (ldb; Nb = 0: unused); the batch rows are split into `chunks` partial sums
 *                             in `workspace` (chunks * (M*N + M) floats) that a second launch adds up in chunk order. */
/* kr_mlp3_forward_shadow with the tiles of each layer split over `waves` (2 or 4) wavefronts of a workgroup: the same LDS-free
 * launch at ~1/waves of the latency (one wave per 16 rows is a serial chain of ~1400 MFMAs: 87 us whatever the batch).  The
 * waves exchange layer 1's output through global memory: h1_out when the caller keeps it, else `scratch`; scratch_floats >=
 * (h1_out ? 0 : ceil(n/16)*16*h1) + ceil(n/16)*waves*64.  Layers 1 and 2 are bitwise those of kr_mlp3_forward_shadow, layer 3's
 * sum is associated per wave (deterministic). */
int kr_mlp3_forward_split(int32_t n, int32_t in_a, int32_t in_b, int32_t h1, int32_t h2, int32_t out_dim, const float *xa, int32_t lda,
                          const float *xb, int32_t ldb, const float *W1, const float *b1, const float *W2, const float *b2,
                          const float *W3, const float *b3, int32_t act, float scale, float *out, float *h1_out, float *h2_out,
                          float *scratch, int64_t scratch_floats, int32_t waves, void *stream);

int kr_mlp3_backward_shadow(int32_t n, int32_t in_dim, int32_t h1, int32_t h2, int32_t out_dim, const float *dz3, const float *W3,
                            const float *h2a, const float *W2, const float *h1a, float *dz2_out, float *dz1_out, const float *W1,
                            int32_t col0, int32_t ncol, const float *act_out, float scale, float *dx_out, void *stream);

/* kr_mlp3_backward_shadow with the dz1 tiles split over `waves` (2 or 4) wavefronts of a workgroup (as kr_mlp3_forward_split);
 * scratch (only used for dx_out: >= ceil(n/16) * waves * 64 floats) holds the waves' partial sums of dx, added in wave order. */
int kr_mlp3_backward_split(int32_t n, int32_t in_dim, int32_t h1, int32_t h2, int32_t out_dim, const float *dz3, const float *W3,
                           const float *h2a, const float *W2, const float *h1a, float *dz2_out, float *dz1_out, const float *W1, int32_t col0,
                           int32_t ncol, const float *act_out, float scale, float *dx_out, float *scratch, int64_t scratch_floats,
                           int32_t waves, void *stream);
int kr_weight_grad_shadow(int32_t n, int32_t M, int32_t Na, int32_t Nb, const float *dz, const float *ha, int32_t lda, const float *hb,
                          int32_t ldb, int32_t chunks, float *workspace, float *dW, float *db, void *stream);

/* Actor forward + exploration noise + kr_select_action in ONE launch (main_DDPGfD.py:424-451): the epilogue of the
 * fused MLP applies the selection rule to its own output.  obs .. ready and action .. lifting as in kr_select_action;
 * W1 .. b3 the actor (82 -> h1 -> h2 -> 4).  Noise: either `noise` [n,4] ~ N(0,1) (then rng_state = NULL), or
 * noise = NULL and rng_state = int64[2] on the device, zero-initialised by the caller: the kernel draws
 * Philox4x32-10 / Box-Muller normals keyed by (seed, rng_state[0], env) and advances rng_state[0] by one per launch
 * (rng_state[1] is its scratch word) - no host-side generator state, so the launch replays from a HIP graph.
 * actor_out [n,4] (optional, may be NULL) receives pi(obs). */
int kr_actor_select(int32_t n, int32_t h1, int32_t h2, const float *obs, const float *prev_obs, const uint8_t *has_prev, const int64_t *t,
                    uint8_t *ready, const float *W1, const float *b1, const float *W2, const float *b2, const float *W3, const float *b3,
                    const float *noise, uint64_t seed, int64_t *rng_state, float sigma, float max_action, int32_t skip_steps,
                    float *actor_out, float *action, float *action_t, uint8_t *lifting, void *stream);

/* ---- the learner's one exchange step (gradient mean over the env-shard ranks; the reference is single-process,
 * its multi-GPU form is DDPGfD.train_batch on every rank + an all-reduce of the gradients, DDPGfD.py:330-356).
 * An LDS-free all-reduce over peer-mapped device memory (csrc/ks_xchg.hip): it runs beside ks_step's stepping kernel, which
 * holds every CU's LDS; a library (RCCL) collective would wait for it.  One process per GPU:
 *   kr_xchg_create   allocates this rank's exchange block for buffers of up to max_count floats and returns its
 *                    KR_XCHG_HANDLE_BYTES-byte inter-process handle (hipIpcGetMemHandle);
 *   kr_xchg_connect  takes the handles of ALL ranks (world x KR_XCHG_HANDLE_BYTES, rank order; gathered by the caller, e.g.
 *                    torch.distributed.all_gather_object) and maps the peers' blocks;
 *   kr_xchg_allreduce_mean  grad[i] <- mean over ranks of grad[i], in place, asynchronous on `stream`; every rank must make
 *                    the same sequence of calls (same counts).  The sum runs in rank order on every rank: bitwise identical
 *                    results everywhere.  grad must be 16-byte aligned;
 *   kr_xchg_status   failed_epoch = 0, or the number of the first call in which a peer did not arrive within ~4 s (the call
 *                    then leaves grad unreduced instead of hanging the GPU). */
#define KR_XCHG_MAX_RANKS 8
#define KR_XCHG_HANDLE_BYTES 64
typedef struct kr_xchg kr_xchg;
int kr_xchg_create(kr_xchg **out, int32_t world, int32_t rank, int64_t max_count, uint8_t *handle_out);
int kr_xchg_connect(kr_xchg *x, const uint8_t *handles);
int kr_xchg_allreduce_mean(kr_xchg *x, float *grad, int64_t count, void *stream);
int kr_xchg_status(kr_xchg *x, uint32_t *failed_epoch);
void kr_xchg_destroy(kr_xchg *x);

#ifdef __cplusplus
}
#endif
#endif
