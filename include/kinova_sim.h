/* include/kinova_sim.h -- C ABI of libkinova_sim.so, the MI355X-native batched replacement for the
 * mujoco-py engine object that KinovaGripper_Env drives one env at a time.
 *
 * Reference interface each entry point replaces (reference root = /root/reference/, file
 * gym-kinova-gripper/gym_kinova_gripper/envs/kinova_gripper_env.py, "ENV"):
 *
 *   ks_create / ks_destroy   <- MjSim(model) construction / garbage collection      ENV:102, 879, 1003
 *   ks_load_model            <- mujoco_py.load_model_from_path(xml)                 ENV:62, 878, 1002
 *                               (the MJCF+STL compile happens offline: model_compiler.py -> .ksm)
 *   ks_load_models /
 *   ks_reset_objects         <- the per-episode object choice: select_object + load of another XML   ENV:986-1005, 1180-1222
 *   ks_reset                 <- KinovaGripper_Env.reset: write_xml (hand euler) + _set_state
 *                               + sim.forward() + _get_obs()                        ENV:1310-1410, 851-881, 692-703
 *   ks_step                  <- KinovaGripper_Env.step: action->ctrl, 15 x sim.step(),
 *                               _get_obs(), _get_reward()                           ENV:1495-1552
 *                               + gym TimeLimit (max_episode_steps)                 gym_kinova_gripper/__init__.py:3-7, main_DDPGfD.py:384
 *   ks_set_env_params        <- (none: the reference edits geom mass / pair friction in the XML; config-5 extension)
 *   ks_obs_from_snapshot     <- _get_obs() + _get_reward() on a given engine state          ENV:438-534, 631-687
 *   ks_get_state/ks_set_state<- sim.data.qpos / qvel / qacc_warmstart, sim.data.ncon,
 *                               contact forces (parity taps)                        ENV:109, 347-353
 *
 * Conventions
 *   - All pointers are DEVICE pointers unless the name ends in _host.  The caller (PyTorch) owns
 *     every buffer; the library owns only its context and scratch.  Pointers are borrowed for the
 *     duration of the stream operation.
 *   - Real-valued buffers have the context precision: float for precision 32 (the product), double
 *     for precision 64 (algorithm-exactness checks only).
 *   - Batched arrays are struct-of-arrays, field-major: x[k * n_envs + env].  The observation may be
 *     requested env-major (obs[env * 82 + k]) with obs_env_major = 1.
 *   - Every call is asynchronous on `stream` (a hipStream_t passed as void*); no hidden host syncs.
 *   - One context per GPU per process; a context is not thread safe.
 *   - Return value: 0 on success, negative ks_status otherwise; ks_last_error gives a message.
 *   - There is NO CPU path: ks_create fails when no HIP device is available.
 *   - Two builds of this ABI exist.  libkinova_sim.so: the reference's fixed topology (ground, 7 hand geoms, ONE object geom) - every
 *     single-geom object, the fast path.  libkinova_sim_mg.so: the same sources with the capacities of the multi-geom objects (the
 *     Bottle / TBottle / Bowl / RBowl models, `object` + welded child bodies: kinova_description/j2s7s300_end_effector_v1_sbottle.xml
 *     :158-186, shape keys ENV:189-208); it also loads single-geom models (a curriculum stage mixes both), more slowly.  A caller binds
 *     the library its models need: ks_load_model of libkinova_sim.so refuses a multi-geom blob with a message that says so.
 */
#ifndef KINOVA_SIM_H
#define KINOVA_SIM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KS_NQ 16
#define KS_NV 15
#define KS_NACT 4
#define KS_NOBS 82
#define KS_NINFO 3
#define KS_NCON_MAX 24      /* libkinova_sim.so; the multi-geom build (libkinova_sim_mg.so) keeps KS_NCON_MAX_MG */
#define KS_NCON_MAX_MG 40
#define KS_CONTACT_STRIDE 20

typedef enum {
    KS_OK = 0,
    KS_ERR_INVALID = -1,  /* bad argument */
    KS_ERR_NO_DEVICE = -2,
    KS_ERR_HIP = -3,      /* a HIP runtime call failed */
    KS_ERR_MODEL = -4,    /* malformed or unsupported model blob */
    KS_ERR_STATE = -5     /* call sequence error (e.g. step before load_model) */
} ks_status;

typedef struct {
    int32_t n_envs;
    int32_t frame_skip;         /* 15 (ENV:51) */
    int32_t horizon;            /* episode time limit in env steps, 30 (main_DDPGfD.py:384); <= 0 disables */
    int32_t solver_iterations;  /* Newton iterations per substep */
    int32_t precision;          /* 32 or 64 */
    int32_t auto_reset;         /* 1: envs that finish are reset to their stored initial state inside ks_step */
    int32_t obs_env_major;      /* 0: obs[k*N+env], 1: obs[env*82+k] */
    int32_t envs_per_wave;      /* envs per stepping workgroup (256 threads = 16 lanes per env): 0 = automatic (16,
                                   or fewer when the model's hull tables leave less LDS); else 1..16 */
    int32_t contact_tap;        /* 1: keep the per-contact records of the last substep for ks_get_state (parity) */
    int32_t pair_memory;        /* 1 (default): every env carries what its hull pairs remember of their last narrow-phase
                                   queries (support hints, closest-feature simplex, penetration portal: vertex ids only)
                                   from one ks_step to the next, so no substep starts its queries cold; 0: cold start
                                   at every ks_step.  Same contacts to the queries' 1e-6 tolerance either way. */
    int32_t reserved[2];
} ks_config;

typedef struct ks_ctx ks_ctx;

void ks_default_config(ks_config *cfg);
int ks_create(const ks_config *cfg, int device, ks_ctx **out);
void ks_destroy(ks_ctx *ctx);
const char *ks_last_error(const ks_ctx *ctx);   /* ctx may be NULL: last creation error */

/* blob_host: KSMB model blob in HOST memory (copied; may be freed after the call).  A context loads its model(s) once. */
int ks_load_model(ks_ctx *ctx, const void *blob_host, size_t nbytes);

/* Mixed-object batches (BASELINE config 5; the reference swaps the object by loading another MJCF per episode,
 * ENV:986-1005): n_models (<= 32) blobs of the SAME hand with different objects; object k of ks_reset_objects is
 * blobs_host[k].  One context, one stepping launch: envs are grouped by object inside the library (every stepping
 * workgroup stages one object's hull tables), the hand's meshes are kept once. */
int ks_load_models(ks_ctx *ctx, int32_t n_models, const void *const *blobs_host, const size_t *nbytes);

/* Reset `n` envs.  env_ids: device int32[n] or NULL for envs 0..n-1 (then n must be n_envs).
 * qpos0: [16, n] start configuration (3 slides, 6 finger joints, object xyz + quat wxyz),
 * hand_quat: [4, n] orientation of j2s7s300_link_7 (the euler the reference patches into the XML).
 * Both are stored as the env's initial state for auto-reset.  obs (optional): observation buffer of
 * the WHOLE batch (layout per cfg); only the rows of the reset envs are written. */
int ks_reset(ks_ctx *ctx, const int32_t *env_ids, int32_t n, const void *qpos0, const void *hand_quat, void *obs, void *stream);

/* ks_reset that also chooses every reset env's object and (optionally) its randomised parameters - the reference's
 * reset(): select_object (ENV:986-1005) + select_orienation (ENV:1180-1222) + _set_state (ENV:692-703).
 * object_id: device int32 [n], index into the blobs of ks_load_models, or NULL (objects stay);
 * mass_friction: device [2, n] (row 0 object mass in kg - the inertia scales with it -, row 1 friction of the seven
 * object-hand pairs), or NULL: an env whose object is (re)assigned takes that object's compiled mass / friction, other
 * envs keep theirs.  (Mass / friction randomisation is an extension: the reference fixes 0.1 kg, XML:153, and mu 1,
 * XML:160-166.)  With object_id in a context of several models the call regroups the stepping kernels' work list and - round 6, unless the
 * stream is being captured - WAITS for the stream once to read back how many 16-env groups hold envs (what ks_rollout schedules: ks_rollout_plan). */
int ks_reset_objects(ks_ctx *ctx, const int32_t *env_ids, int32_t n, const void *qpos0, const void *hand_quat, const int32_t *object_id,
                     const void *mass_friction, void *obs, void *stream);

/* One env.step() for every env.  action: [4, N] (wrist, finger1..3), obs: N x 82, reward: [N],
 * done: uint8 [N] (bit0 lifted, bit1 time limit), info: [3, N] (finger, grasp, lift reward).
 * final_obs (optional): with auto_reset, rows of envs that finished hold their terminal
 * observation while `obs` already holds the observation after the reset.
 * Stream capture: ks_step / ks_rollout may be recorded into a HIP graph.  The library then records a host-to-device copy
 * of the call's output-pointer record from pinned memory it owns for the life of the context (one record per captured call,
 * never recycled), so any number of captured graphs with different output buffers can be replayed in any order; the
 * buffers themselves are borrowed for as long as the graph may be replayed.  Eager calls may be mixed with replays: once a context
 * holds a captured call, every eager ks_step re-sends its own output record (the "same pointers as the last call" shortcut is off). */
int ks_step(ks_ctx *ctx, const void *action, void *obs, void *reward, uint8_t *done, void *info, void *final_obs, void *stream);

/* Parity taps; any pointer may be NULL.  contact: [KS_NCON_MAX*KS_CONTACT_STRIDE, N] (libkinova_sim_mg.so: KS_NCON_MAX_MG) records of the
 * last substep (pos3 normal3 dist mu bodies+pair (b1 + 16 b2 + 256 pair index) R aref4 force3(normal,t1,t2) active-row-mask D spare; needs cfg.contact_tap), ncon: int32 [N],
 * status: int32 [N] sticky bit flags (1 contact overflow, 2 non-finite state, 4 the env's rays were not delivered in time by the
 * stepping launch's ray pool - a wait ran out; never seen in practice, reported instead of hanging; 8 a substep's Newton iteration
 * ended at cfg.solver_iterations before its stop rule fired: that substep used a truncated iterate). */
int ks_get_state(ks_ctx *ctx, void *qpos, void *qvel, void *qacc_warmstart, void *contact, int32_t *ncon, int32_t *status, void *stream);
int ks_set_state(ks_ctx *ctx, const void *qpos, const void *qvel, const void *qacc_warmstart, void *stream);

/* Per-env domain randomisation (BASELINE config 5; an extension: the reference fixes the object's mass at 0.1 kg,
 * XML:153, and the object-hand friction at 1, XML:160-166).  obj_mass [N] (kg; the inertia scales with it), obj_mu [N]
 * (friction of the seven object-hand pairs); either pointer may be NULL = leave as is.  Defaults are the model's. */
int ks_set_env_params(ks_ctx *ctx, const void *obj_mass, const void *obj_mu, void *stream);

/* Advance by ONE mj_step with explicit controls ctrl [9, N] (no observation); parity testing. */
int ks_substep(ks_ctx *ctx, const void *ctrl, void *stream);

/* Observation / reward / termination of caller-provided kinematic snapshots: snap [105, N] = world poses of bodies 2..9
 * (j2s7s300_link_7, finger 1 proximal, distal, finger 2 ..., object; 12 values each: rotation row-major 9, position 3)
 * followed by the 9 jointpos sensors (slides, proximal 1-3, distal 1-3); rays [17, N] = the rangefinder distances
 * (-1 = no hit).  Replaces _get_obs() + _get_reward() on a given mujoco-py state (ENV:438-534, 631-687) without any
 * stepping: done bit0 = lifted; no step counter, no time limit, no auto-reset.  Overwrites the context's snapshot and ray
 * buffers (call ks_reset afterwards to continue an episode).  Parity hook: the reference-generated env-layer golden
 * vectors go through the HIP observation kernel this way. */
int ks_obs_from_snapshot(ks_ctx *ctx, const void *snap, const void *rays, void *obs, void *reward, uint8_t *done, void *info, void *stream);

/* Free-running rollout: n_iter env-steps of EVERY env in ONE launch, every stepping workgroup (16 envs, one CU) running its own
 * loop with no synchronisation between workgroups -
 *     actor forward of its envs (3-layer MLP on the matrix cores, newest published weights) + exploration noise + the
 *     check_grasp / scripted-lift rule  ->  15 substeps  ->  rangefinder rays  ->  observation / reward / done / auto-reset
 *     ->  replay write (open-episode buffers) + episode hand-over
 * i.e. what kr_actor_select + ks_step + kr_store_transition do per env-step (main_DDPGfD.py:424-464 around ENV:1495-1552), fused.
 * A lock-step launch lasts as long as its slowest wave (1.6 - 1.9 x the median); here a wave (round 6: with one 16-env group per
 * workgroup the four waves of a workgroup never meet inside a launch; otherwise a workgroup) starts its next env-step the
 * moment it has finished the last.  Per env the arithmetic, the noise stream (Philox keyed by seed, the env's own step count,
 * env) and therefore the trajectory are those of the lock-step calls for the same weights.
 * A context with more 16-env groups than the GPU has compute units (8192 envs; many objects: one partly filled group per object)
 * keeps one persistent workgroup per compute unit; the groups are taken from a ready queue - a workgroup steps the group at its
 * head once and puts it back behind the others - so that no workgroup idles while a group is ready and every env still does
 * exactly n_iter env-steps per launch (environment variable KS_ROLLOUT_DEAL=static / rr: fixed deals instead; the multi-geom
 * build keeps the round-robin deal).
 *
 * Actor weights: `actor_pub` holds 3 parameter buffers of `actor_stride` floats (layout: offsets off_*, torch.nn.Linear
 * layout); `actor_ver` is a device counter that the learner increments AFTER it has completely written buffer
 * (ver % 3): a workgroup reads the counter at the start of every env-step and uses that buffer (and repeats the forward in the
 * unlikely case that two more versions were published meanwhile).
 * Episodes: two open-episode buffers per env, cur_* [2][n][horizon][...]; a finished episode that is kept (len - n_steps > 1,
 * main_DDPGfD.py:469-471) is handed over by setting pub_len[buf][env] = len (release) and switching to the other buffer; the
 * consumer (the learner's stream: kr_rank / kr_commit / kr_advance_ring on buffer `buf` with cur_len = pub_len + buf * n) clears
 * it.  If the other buffer has not been consumed yet the finished episode is dropped and counted in `dropped`.
 * fp32 contexts with observations in the stepping kernel only (the default); obs layout env-major; one model or many. */
typedef struct {
    const float *actor_pub;
    const int64_t *actor_ver;
    int64_t actor_stride;
    int64_t off_w1, off_b1, off_w2, off_b2, off_w3, off_b3;
    int32_t h1, h2;                  /* hidden widths: 256-256, 400-300, 128-128 or 64-64 */
    float sigma, max_action;         /* exploration noise std (max_action * expl_noise), action bound */
    int32_t skip_steps;              /* check_grasp from this step of an episode on (6, main_DDPGfD.py:418) */
    int32_t with_replay;
    uint64_t seed;
    /* engine state (kinovagrasping_amd.rollout.RolloutEngine) */
    float *obs, *prev_obs;           /* [n, 82] */
    uint8_t *has_prev, *ready, *lifting;
    int64_t *t, *steps_total;        /* [n] steps in the episode / since the start (the noise counter; stored device-visibly: kr_wait_min paces the learner's stream on it) */
    float *action, *action_t;        /* [n, 4], [4, n] */
    float *reward_out; uint8_t *done_out;
    /* ks_step's outputs (as passed to ks_step) */
    float *sim_obs, *sim_reward; uint8_t *sim_done; float *sim_info, *sim_final_obs;
    /* open episodes */
    int32_t horizon, n_steps;
    float *cur_state, *cur_next, *cur_action, *cur_reward, *cur_not_done;
    int64_t *cur_len;                /* [2][n] */
    uint8_t *cur_sel;                /* [n] */
    int64_t *pub_len;                /* [2][n] */
    int64_t *counters;               /* [4]: episodes finished, lifted, kept, dropped (a -DKS_ROLLOUT_STAMP diagnostic build writes [8 + 4 * 512 + 8]) */
    /* Round 6, opt-in: 0 = every env does exactly n_iter env-steps (the default and what every parity test runs).  > 0: a TIME budget in ticks of the
     * 100 MHz device clock - a wave starts no further env-step of its four envs once that long has passed since it entered the launch, so envs advance
     * by time: between 1 and n_iter env-steps each, steps_total[] says how many.  Per env the trajectory is still the lock-step one (same arithmetic, noise
     * keyed by the env's own step count).  For contexts whose objects differ widely in cost (a curriculum stage holding bowls and cubes: the slowest
     * object otherwise paces every env) - at the price of an uneven object mix in what is collected.  Needs the wave form (ks_rollout_plan:
     * KS_PLAN_WAVES), KS_ERR_STATE otherwise. */
    int64_t budget_ticks;
} ks_rollout_args;
int ks_rollout(ks_ctx *ctx, int32_t n_iter, const ks_rollout_args *args_host, void *stream);

/* How ks_rollout would schedule this context's env groups (16 envs sharing one object's hull tables) on the device it runs on - the library's
 * own decision, for callers that report or warn about it (bench.py, pipeline.AsyncTrainer; ADVICE r5: they used to re-derive it from environment
 * variables).  *mode: KS_PLAN_WAVES one group per persistent workgroup, its four waves run free (no barrier inside a launch);
 * KS_PLAN_WORKGROUPS one group per workgroup, the waves joined by barriers at every phase (KS_ROLLOUT_WAVES=0, or a context the wave form cannot
 * hold); KS_PLAN_QUEUE more groups than resident workgroups, taken from a ready queue; KS_PLAN_RUNS / KS_PLAN_ROUND_ROBIN more groups than
 * workgroups in a fixed deal (contiguous runs / round-robin: KS_ROLLOUT_DEAL=static / rr, and the multi-geom library's default) - a launch then
 * runs at the pace of the workgroup with the most groups.  *groups: 16-env groups incl. the partly filled one of every object;
 * *workgroups: persistent workgroups of a launch.  After ks_load_model(s). */
enum { KS_PLAN_WAVES = 0, KS_PLAN_WORKGROUPS = 1, KS_PLAN_QUEUE = 2, KS_PLAN_RUNS = 3, KS_PLAN_ROUND_ROBIN = 4 };
int ks_rollout_plan(ks_ctx *ctx, int32_t *mode, int32_t *groups, int32_t *workgroups);

/* HIP event timing of the dominant kernel: average duration (ms) of the env-step kernel launches
 * since the last call with reset != 0, measured with hipEvents on the launch stream - every 4th launch is sampled (the two
 * event records cost ~8 us of stream time per sampled launch; KS_EVENT_STRIDE=1 samples all of them).  Host sync.
 * (fp32 contexts: that kernel is the 15 substeps of every env plus the rangefinder rays of the workgroup's own envs.) */
int ks_kernel_time(ks_ctx *ctx, int reset, double *avg_ms_host, int64_t *launches_host);

int ks_version(void);

#ifdef __cplusplus
}
#endif
#endif
