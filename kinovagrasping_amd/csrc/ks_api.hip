// ks_api.hip -- gfx950 kernels + C ABI of libkinova_sim.so (see include/kinova_sim.h).
//
// Kernels:
//   k_env_step   15 x mj_step per launch: 16 lanes per env, 4 envs per wave, 16 envs per 256-thread workgroup, all
//                per-env data + model + hull tables in LDS (~155 KB).  THE dominant kernel.  fp32 contexts: each
//                workgroup then casts the 17 rangefinder rays of its own envs before it retires (wg_rays).
//   k_obs        82-d observation, reward, termination, time limit; on an auto-reset returns the cached observation of
//                the env's stored initial state and restarts the env (one env per lane)
//   k_reset      state <- stored initial state for flagged envs, kinematics -> snapshot (ks_reset only)
//   k_rays       17 rangefinder rays x 8 mesh geoms as a launch of their own, one (env, ray, geom) per lane, grid
//                (N/8, 17): ks_reset, fp64 contexts, KS_RAYS_IN_STEP=0
//   k_substep    one mj_step with explicit controls (parity tap)
// No CPU fallback exists in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <map>
#include <string>
#include <vector>

#include "../../include/kinova_sim.h"
#include "ks_env.h"
#include "ks_mlp_tile.h"
#include "ks_select.h"
#include "ks_model_host.h"

using namespace ks;

namespace {

constexpr int WAVE = 64;
thread_local std::string g_create_error;

template <typename T> struct Buffers {
    T *qpos, *qvel, *warm;        // [16|15|15][N]
    T *hand_quat, *qpos0;         // [4][N], [16][N]   stored initial state
    T *snap;                      // [SNAP_TOTAL][N]
    T *rays;                      // [17][N]
    T *obs0;                      // [N][82] observation of every env's stored initial state (written by the reset pass):
                                  // an auto-reset returns this row instead of casting the rays of the initial state again
    T *contact;                   // [NCON_MAX*CON_STRIDE][N] parity tap (last substep)
    T *gscratch;                  // [SCR_TOTAL][N] (fp64 contexts only; fp32 uses LDS)
    T *envp;                      // [2][N] per-env object mass, object-hand friction (config 5 randomisation)
    unsigned *pairmem;            // [N][SUBS][2 * WARM_WORDS] pair memory carried from launch to launch (fp32 / LDS contexts):
                                  // what every lane of an env's team remembers of its (<= 2) hull pairs' last queries
                                  // (ks_core.h: PairWarm), env-major: a team moves its 512 bytes as 16 x 32-byte pieces
    int32_t *ncon, *status, *step_count;
    uint8_t *flag;                // envs to (re)initialise
    // mixed-object batches (BASELINE config 5): the object model of every env, and the stepping kernel's work list - its
    // workgroups stage ONE object's hull tables, so envs are grouped by object: slot s of the list holds an env id or -1
    // (a group is padded to whole workgroups), workgroup w steps slots [w * epw, (w + 1) * epw) with model wg_model[w]
    int32_t *obj_id;              // [N]
    int32_t *slot_env;            // [n_wg * epw]
    int32_t *wg_model;            // [n_wg]
    int32_t *n_used;              // [1] groups of the list that hold envs (k_slots)
    T *nominal;                   // [n_models][2] object mass / object-hand friction of every model as compiled
    // the ray pool of a stepping launch (wg_ray_pool): [0] tickets published, [1] tickets claimed, [2] workgroups that have left,
    // [3] workgroups that have started,
    // [4 .. 4 + n_wg) the published workgroups in order, [4 + n_wg .. 4 + 2 n_wg) their done flags; all zero between launches
    int32_t *rayq;
};

template <typename T> struct ColW {
    T* base;
    long stride;
    __device__ void operator()(int k, T v) const { base[(long)k * stride] = v; }
};

// Dynamic LDS layout of the stepping kernels: [hull vertex tables][per-lane scratch, lpw lanes].
// Every thread of the (single-wave) workgroup helps to stage the hull tables from global memory; the
// tables are then read with wave-uniform ds_read broadcasts.
template <typename T> struct LdsScratch { using type = Scratch<T, KS_LDS T*>; };

// The model constants (about 3 KB) are copied to the head of LDS as well: the physics stages are out-of-line
// device functions that reach the model through a generic reference, and a flat load that resolves to LDS costs
// a fraction of one that goes to L2.  Returns the words (of T) used; visibility comes with stage_hulls' barrier.
constexpr int HULLS_BYTES = MULTI_GEOM ? 1024 : 256;   // the Hulls descriptor (table pointers, counts) sits behind the model copy
template <typename T> constexpr int model_words() { return (int)(((sizeof(Model<T>) + 15) / 16 * 16 + HULLS_BYTES) / sizeof(T)); }
template <typename T> __device__ __forceinline__ const Model<T>* stage_model(const Model<T>* __restrict__ mp, KS_LDS T* lds) {
    const unsigned* src = (const unsigned*)mp;
    KS_LDS unsigned* dst = (KS_LDS unsigned*)lds;
    for (int i = threadIdx.x; i < (int)(sizeof(Model<T>) / 4); i += blockDim.x) dst[i] = src[i];
    return (const Model<T>*)lds;
}

// size of a PairRec in DEVICE code (LDS pointers are 4 bytes there; the host pass of this file sees 8)
template <typename T> constexpr int pair_rec_bytes() { return MULTI_GEOM ? (sizeof(T) == 4 ? 128 : 160) : (sizeof(T) == 4 ? 96 : 144); }   // (multi-geom: generic table pointers, 8 bytes)

// Global -> LDS copy of a table by the whole workgroup in B-byte words (both ends B-byte aligned, `bytes` a multiple of B), four
// loads in flight per thread: the tables of a workgroup are ~50 KB, and copied element by element (floats, 16-bit ids) the
// staging was 56 dependent load -> store rounds = 27 us at the head of every launch.
template <int B, typename D, typename Sx> __device__ __forceinline__ void stage_copy(KS_LDS D* dst, const Sx* src, size_t bytes) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    using W = typename std::conditional<B == 16, v4u, v2u>::type;
    KS_LDS W* d = (KS_LDS W*)dst;
    const W* s = (const W*)src;
    const int n = (int)(bytes / B), step = blockDim.x;
    int i = threadIdx.x;
    for (; i + 3 * step < n; i += 4 * step) {
        const W a = s[i], b = s[i + step], c = s[i + 2 * step], e = s[i + 3 * step];
        d[i] = a; d[i + step] = b; d[i + 2 * step] = c; d[i + 3 * step] = e;
    }
    for (; i < n; i += step) d[i] = s[i];
}

// The packed hull tables (Model::hull_pack, already in LDS order) in one pass: up to eight 16-byte loads in flight per thread, so
// the ~50 KB of a workgroup are two round trips to the L2 instead of twelve table-by-table copies (each a round trip or more).
template <typename D> __device__ __forceinline__ void stage_tables(KS_LDS D* dst, const void* src, int bytes) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    KS_LDS v4u* d = (KS_LDS v4u*)dst;
    const __attribute__((address_space(1))) v4u* s = (const __attribute__((address_space(1))) v4u*)src;
    const int n = bytes >> 4, step = blockDim.x;
    constexpr int K = 8;
    // (every load unconditional, with a clamped index: a register array that is only conditionally loaded makes the compiler
    // wait for each load on its own)
    for (int i = threadIdx.x; i < n; i += K * step) {
        v4u r[K];
        KS_UNROLL
        for (int k = 0; k < K; k++) r[k] = s[i + k * step < n ? i + k * step : n - 1];
        KS_UNROLL
        for (int k = 0; k < K; k++)
            if (i + k * step < n) d[i + k * step] = r[k];
    }
}

// Model constants and packed hull tables of a workgroup in one go: the model's loads and the first sixteen table loads of every
// thread are in flight together (the table staging used to start behind the model copy and a barrier, because it read its
// pointers from the LDS copy).  Returns the model in LDS; the tables land behind it, where stage_hulls expects them.
template <typename T, int NT> __device__ __forceinline__ const Model<T>* stage_model_and_tables(const Model<T>* __restrict__ mp, KS_LDS T* lds) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    constexpr int MW = (int)(sizeof(Model<T>) / 4), MR = (MW + NT - 1) / NT, K = 8;
    const int tid = threadIdx.x;
    const __attribute__((address_space(1))) v4u* ps = (const __attribute__((address_space(1))) v4u*)mp->hull_pack;
    const int n = mp->hull_pack_bytes >> 4;
    const __attribute__((address_space(1))) unsigned* src = (const __attribute__((address_space(1))) unsigned*)mp;
    KS_LDS unsigned* dm = (KS_LDS unsigned*)lds;
    KS_LDS v4u* dt = (KS_LDS v4u*)(lds + model_words<T>());
    unsigned mr[MR];
    v4u r[K], r2[K];
    auto at = [&](int j) { return j < n ? j : n - 1; };
    // unconditional loads with clamped indices (see stage_tables); a context always has the pack (n >= 1).  The tables of a
    // workgroup are ~50 KB = 12.5 loads per thread: two sets of eight in flight together cover them in one round trip
    const bool more = n > K * NT;                 // (uniform)
    KS_UNROLL
    for (int k = 0; k < MR; k++) mr[k] = src[tid + k * NT < MW ? tid + k * NT : MW - 1];
    if (n > 0) {
        KS_UNROLL
        for (int k = 0; k < K; k++) r[k] = ps[at(tid + k * NT)];
        if (more) {
            KS_UNROLL
            for (int k = 0; k < K; k++) r2[k] = ps[at(tid + (K + k) * NT)];
        }
    }
    KS_UNROLL
    for (int k = 0; k < MR; k++)
        if (tid + k * NT < MW) dm[tid + k * NT] = mr[k];
    if (n > 0) {
        KS_UNROLL
        for (int k = 0; k < K; k++)
            if (tid + k * NT < n) dt[tid + k * NT] = r[k];
        if (more) {
            KS_UNROLL
            for (int k = 0; k < K; k++)
                if (tid + (K + k) * NT < n) dt[tid + (K + k) * NT] = r2[k];
        }
    }
    for (int i = tid + 2 * K * NT; i < n; i += K * NT) {
        KS_UNROLL
        for (int k = 0; k < K; k++) r[k] = ps[at(i + k * NT)];
        KS_UNROLL
        for (int k = 0; k < K; k++)
            if (i + k * NT < n) dt[i + k * NT] = r[k];
    }
    return (const Model<T>*)lds;
}

// SHARED: `hu` is the workgroup's one descriptor in LDS, filled by thread 0 (every thread writing its own private copy
// was 240 bytes of stack per lane, written through to memory at every launch: 16 of the 27 MB a launch wrote);
// otherwise `hu` is the calling thread's own.
template <typename T, bool SHARED> __device__ __forceinline__ void stage_hulls(const Model<T>& m, KS_LDS T* lds, int& used, Hulls<T>& hu, bool prestaged = false) {
    const bool writer = !SHARED || threadIdx.x == 0;
#ifdef KS_MULTI_GEOM
    // multi-geom build: the hull tables stay in global memory (ks_math.h: KS_TAB) - the descriptor points at the model's arrays
    (void)prestaged;
    if (writer)
        for (int s = 0; s < NMESH; s++) {
            hu.vert[s] = m.mesh_vert[s]; hu.nvert[s] = m.mesh_nvert[s]; hu.nvert_pad[s] = m.mesh_nvert_pad[s];
            hu.adj_off[s] = m.mesh_adj_off[s]; hu.adj[s] = m.mesh_adj[s];
        }
    used = 0;
#else
    const bool packed = m.hull_pack != nullptr;
    if (packed && !prestaged) stage_tables(lds, m.hull_pack, m.hull_pack_bytes);
    int off = 0;
    for (int s = 0; s < 4; s++) {
        const int n = m.mesh_nvert_pad[s] * 4;
        const T* src = m.mesh_vert[s];
        if (!packed) stage_copy<16>(lds + off, src, (size_t)n * sizeof(T));
        if (writer) {
            hu.vert[s] = lds + off;
            hu.nvert[s] = m.mesh_nvert[s];
            hu.nvert_pad[s] = m.mesh_nvert_pad[s];
        }
        off += n;
    }
    // adjacency (uint16 chunk tables) behind the vertex tables; `used` stays in units of T
    KS_LDS unsigned short* ulds = (KS_LDS unsigned short*)(lds + off);
    int uoff = 0;
    for (int s = 0; s < 4; s++) {
        const int no = m.mesh_nvert[s] + 1, na = m.mesh_nchunk[s] * 4;
        if (!packed) stage_copy<8>(ulds + uoff, m.mesh_adj_off[s], (size_t)((no + 3) & ~3) * sizeof(unsigned short));   // (padded to whole 8-byte words at load)
        if (writer) hu.adj_off[s] = ulds + uoff;
        uoff += (no + 3) & ~3;                       // keep the chunk tables 8-byte aligned
        if (!packed) stage_copy<8>(ulds + uoff, m.mesh_adj[s], (size_t)na * sizeof(unsigned short));
        if (writer) hu.adj[s] = ulds + uoff;
        uoff += na;
    }
    const int iwords = (uoff * (int)sizeof(unsigned short) + (int)sizeof(T) - 1) / (int)sizeof(T);
    used = off + ((iwords + 3) & ~3);
#endif
    // pair records behind the adjacency tables, one thread per pair
    if (writer) {
        hulls_set_pairs(m, hu);
        hu.pair = (KS_LDS const PairRec<T>*)(lds + used);
    }
    if (SHARED) __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(sizeof(PairRec<T>) == pair_rec_bytes<T>(), "device size of a pair record");
#endif
    for (int pi = threadIdx.x; pi < m.npair; pi += blockDim.x) fill_pair_rec(m, hu, pi, *(PairRec<T>*)((KS_LDS const PairRec<T>*)(lds + used) + pi));
    used += NPAIR_MAX * pair_rec_bytes<T>() / (int)sizeof(T);
    __syncthreads();
}

template <typename T> __device__ __forceinline__ void load_state(const Buffers<T>& b, int env, int N, LaneState<T>& st) {
    KS_UNROLL
    for (int i = 0; i < NQ; i++) st.qpos[i] = b.qpos[(long)i * N + env];
    KS_UNROLL
    for (int i = 0; i < NV; i++) { st.qvel[i] = b.qvel[(long)i * N + env]; st.warm[i] = b.warm[(long)i * N + env]; }
}
template <typename T> __device__ __forceinline__ void store_state(const Buffers<T>& b, int env, int N, const LaneState<T>& st) {
    KS_UNROLL
    for (int i = 0; i < NQ; i++) b.qpos[(long)i * N + env] = st.qpos[i];
    KS_UNROLL
    for (int i = 0; i < NV; i++) { b.qvel[(long)i * N + env] = st.qvel[i]; b.warm[(long)i * N + env] = st.warm[i]; }
}

// team versions: the env state lives in the env's LDS block, every lane moves a share of the 46 values
template <typename T, int SUBS_> __device__ __forceinline__ void load_state_team(const Buffers<T>& b, int env, int N, T* st, int sub) {
    for (int i = sub; i < NQ + 2 * NV; i += SUBS_)
        st[i] = i < NQ ? b.qpos[(long)i * N + env] : (i < NQ + NV ? b.qvel[(long)(i - NQ) * N + env] : b.warm[(long)(i - NQ - NV) * N + env]);
}
template <typename T, int SUBS_> __device__ __forceinline__ void store_state_team(const Buffers<T>& b, int env, int N, const T* st, int sub) {
    for (int i = sub; i < NQ + 2 * NV; i += SUBS_) {
        if (i < NQ) b.qpos[(long)i * N + env] = st[i];
        else if (i < NQ + NV) b.qvel[(long)(i - NQ) * N + env] = st[i];
        else b.warm[(long)(i - NQ - NV) * N + env] = st[i];
    }
}
static_assert(sizeof(LaneState<float>) == (NQ + 2 * NV) * sizeof(float), "LaneState is qpos | qvel | warm, packed");

// per-env randomised parameters -> the env's scratch block
template <typename T, typename S, int SUBS_> __device__ __forceinline__ void load_env_params(S scr, Team<SUBS_> team, const Buffers<T>& b, int env, int N) {
    if (team.sub == 0) { scr(SCR_ENVP) = b.envp[env]; scr(SCR_ENVP + 1) = b.envp[(long)N + env]; }
    team.sync();
}

constexpr int SUBS = 16;         // lanes per env (a DPP row)
#ifndef KS_LANE_STRIDE
#define KS_LANE_STRIDE 16
#endif
constexpr int LANE_STRIDE = KS_LANE_STRIDE;   // lanes reserved per env (16 = teams packed; 32 = every second DPP row idle: 2 envs per wave)
constexpr int WG = 16 * LANE_STRIDE;          // stepping workgroup: 16 envs sharing the hull tables in LDS
constexpr int EPW_MAX = WG / LANE_STRIDE;

// ---- the rangefinder rays of a workgroup's own envs, inside the stepping kernel (fp32 / LDS variant).
// A workgroup that has finished its 15 substeps casts the 17 rays of its 16 envs before it retires: workgroups finish
// at different times (0.7 .. 1.1 ms), so all but the last do this while others are still stepping - the separate k_rays
// launch (and its launch gap) leaves the critical path, only the slowest workgroup's own rays stay on it.
// Two passes over the (env, ray, geom) tasks: every thread culls its share against the geoms' bounding boxes and appends
// the survivors (6 - 15 %) to a queue in LDS; the queue is then walked one entry per thread - dense lanes instead of one
// walker per eight - and walkers hand far subtrees to waiting lanes through the same queue (SharingStack below).  The nearest hit of an (env, ray) is an LDS word updated with atomicMin on the float's bit pattern
// (non-negative floats order like unsigned integers); walkers read it as their pruning bound, exactly as k_rays' group
// bound.  The per-env LDS blocks are dead by then and hold the list, the hit words and the traversal stacks.
struct SlotBound {
    KS_LDS unsigned* slot;
    __device__ float operator()(float best) const {
        if (best >= 0) atomicMin((unsigned*)slot, (unsigned)__float_as_int(best));
        return __int_as_float((int)*(volatile KS_LDS unsigned*)slot);
    }
};
constexpr int RG_BITS = MULTI_GEOM ? 4 : 3, RG = 1 << RG_BITS;        // mesh geom slots of an env in the ray kernels (geoms 1 .. RG)
static_assert(RG == NGEOM - 1, "ray tasks: one slot per mesh geom");
constexpr int WG_RAY_TASKS = NRAY * RG;                                // per env
constexpr int WG_SNAP = 97;                                            // body poses of an env (96 floats), odd stride
// the walk queue: every surviving (env, ray, geom) task + the subtrees that busy walkers hand to idle lanes
__host__ __device__ constexpr int wg_ray_queue(int epw) { return epw * WG_RAY_TASKS + 64 * epw; }
__host__ __device__ constexpr int wg_rays_words(int epw, int nth = WG) { return epw * NRAY + 8 + 2 * wg_ray_queue(epw) + RAY_STACK * nth + epw * WG_SNAP; }
// ... followed by what wg_obs adds: the final ray distances [epw][NRAY + 1] and an auto-reset's kinematics scratch [epw][SCR_CON + 1]
__host__ __device__ constexpr int wg_obs_words(int epw) { return epw * (NRAY + 1) + epw * (SCR_CON + 1); }
struct LdsSnap {
    KS_LDS const float* base;
    __device__ float operator()(int k) const { return base[k]; }
};
// A walker's pending subtrees: its own stack in LDS (LdsStack) - or, while lanes of the workgroup wait for work, the shared queue.
// The busiest lane of a workgroup used to visit ~23 nodes while the average lane visits 4.5 (170 surviving tasks on 256 lanes): the
// pass lasted as long as the longest walk.  ctl[0] = entries appended, ctl[1] = tickets taken (a lane takes ONE ticket and waits for
// that entry: appends reach the waiting lanes in ticket order, nothing is polled twice), ctl[2] = walks not yet finished.  An entry is
// two words: the LdsStack word (entry parameter | node) and task + 1, written last (0 = not there yet).  The nearest hit of an (env, ray)
// is shared through its LDS word (SlotBound) anyway, so a subtree walked by another lane prunes and is pruned exactly as before, and
// the minimum over the same set of triangles does not depend on who visits them.
struct SharedWalks {
    KS_LDS unsigned* ctl;
    KS_LDS unsigned* q;
    int cap;
};
struct SharingStack {
    LdsStack<float> own;
    SharedWalks sh;
    unsigned task1;
    __device__ void push(int n, float tt) {
        if (*(volatile KS_LDS unsigned*)(sh.ctl + 1) > *(volatile KS_LDS unsigned*)sh.ctl) {       // somebody holds a ticket for an entry that does not exist yet
            const unsigned i = atomicAdd((unsigned*)sh.ctl, 1u);
            if (i < (unsigned)sh.cap) {
                atomicAdd((unsigned*)(sh.ctl + 2), 1u);
                unsigned b = (unsigned)__float_as_int(tt);
                b = b >= 0x10000u ? b - 0x10000u : 0u;
                sh.q[2 * i] = (b & 0xffff0000u) | (unsigned)n;
                __threadfence_block();
                *(volatile KS_LDS unsigned*)(sh.q + 2 * i + 1) = task1;
                return;
            }
        }
        own.push(n, tt);
    }
    __device__ bool pop(int& n, float& tt) { return own.pop(n, tt); }
};

#ifdef KS_ROLLOUT_STAMP
#define KS_RAY_PROF_PARAM , long long* rprof
#define KS_RAY_PROF_ARG(p) , p
#define KS_RP(i) { const long long t1_ = wall_clock64(); if (rprof && threadIdx.x == 0) atomicAdd((unsigned long long*)&rprof[i], (unsigned long long)(t1_ - rtk)); rtk = t1_; }
#else
#define KS_RAY_PROF_PARAM
#define KS_RAY_PROF_ARG(p)
#define KS_RP(i)
#endif
// The threads that work together on a tail of the stepping kernels: the whole workgroup (k_env_step; k_rollout when its workgroups
// step several groups), or ONE WAVE with its own four envs (k_rollout's free-running waves: no s_barrier anywhere in their loop - the
// wave's LDS and memory operations are ordered by program order plus a fence).
template <bool WAVE_SCOPE> struct Crew {
    static constexpr int NTH = WAVE_SCOPE ? WAVE : WG;
    static __device__ __forceinline__ int tid() { return WAVE_SCOPE ? (int)(threadIdx.x & (WAVE - 1)) : (int)threadIdx.x; }
    static __device__ __forceinline__ void sync() {
        if constexpr (WAVE_SCOPE) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    }
};
template <bool WAVE_SCOPE>
__device__ __noinline__ void wg_rays(const Model<float>& m, const Buffers<float>& b, int N, int slot0, int epw, KS_LDS unsigned* w KS_RAY_PROF_PARAM) {
    using C = Crew<WAVE_SCOPE>;
    constexpr int NTH = C::NTH;
#ifdef KS_ROLLOUT_STAMP
    long long rtk = wall_clock64();
    int my_visits = 0;
#endif
    KS_LDS unsigned* hit = w;                                          // [epw][17] nearest hit so far (float bits), big = none
    KS_LDS unsigned* ctl = w + epw * NRAY;                             // [8]: queue tail, tickets, unfinished walks
    KS_LDS unsigned* q = ctl + 8;                                      // [cap][2] the walk queue
    const int cap = wg_ray_queue(epw);
    KS_LDS unsigned* stk = q + 2 * cap;                                // [RAY_STACK][NTH]
    KS_LDS float* snaps = (KS_LDS float*)(stk + RAY_STACK * NTH);       // [epw][WG_SNAP] body poses (one global read round for everything)
    const int tid = C::tid(), total = epw * WG_RAY_TASKS;
    for (int i = tid; i < epw * NRAY; i += NTH) hit[i] = (unsigned)__float_as_int(Lim<float>::big);
    for (int i = tid; i < cap; i += NTH) q[2 * i + 1] = 0u;
    for (int i = tid; i < epw * 96; i += NTH) {
        const int e = i / 96, k = i % 96, env = b.slot_env[slot0 + e];
        snaps[e * WG_SNAP + k] = env >= 0 ? b.snap[(long)(SNAP_BP + k) * N + env] : 0.f;
    }
    if (tid < 8) ctl[tid] = 0;
    C::sync();
    for (int task = tid; task < total; task += NTH) {
        const int e = task / WG_RAY_TASKS, r = (task % WG_RAY_TASKS) >> RG_BITS, g = 1 + (task & (RG - 1)), env = b.slot_env[slot0 + e];
        if (env < 0 || g >= m.ngeom) continue;
        LdsSnap snap{snaps + e * WG_SNAP - SNAP_BP};
        float pnt[3], vec[3];
        const int sb = ray_origin(m, snap, r, pnt, vec);
        if (g == 1) {
            const float tg = ray_ground(m, pnt, vec);
            if (tg >= 0) atomicMin((unsigned*)(hit + e * NRAY + r), (unsigned)__float_as_int(tg));
        }
        if (m.geom_body[g] != sb && ray_may_hit_geom(m, snap, g, pnt, vec)) {
            const unsigned i = atomicAdd((unsigned*)ctl, 1u);           // (at most `total` <= cap of these)
            q[2 * i] = 0u;                                              // the root, entry parameter 0
            q[2 * i + 1] = (unsigned)task + 1u;
        }
    }
    C::sync();
    if (tid == 0) ctl[2] = ctl[0];
    C::sync();
    KS_RP(0)                             // snapshot load + culling pass
    // The walks, one node visit per loop iteration: a lane whose walk has ended takes the next entry of the queue in the SAME loop, so
    // the lanes of a wave do not wait for the longest walk of a round.
#ifdef KS_ROLLOUT_STAMP
    const unsigned nlist = ctl[0];
#endif
    const SharedWalks sh{ctl, q, cap};
    RayWalk<float, SlotBound, SharingStack> walk;
    KS_LDS unsigned* slot = hit;
    bool busy = false;
    unsigned ticket = 0xffffffffu;
    // ONE loop for the whole wave, left by all of its lanes together when no walk of the workgroup is unfinished (the same LDS word
    // for every lane: a uniform branch).  A lane without work makes one attempt per iteration and falls through - it must never spin
    // in a loop of its own: the entry it waits for may have to be appended by a lane of its own wave.
    while (*(volatile KS_LDS unsigned*)(ctl + 2) != 0u) {
        if (!busy) {
            if (ticket == 0xffffffffu) ticket = atomicAdd((unsigned*)(ctl + 1), 1u);
            const unsigned t1 = ticket < (unsigned)cap ? *(volatile KS_LDS unsigned*)(q + 2 * ticket + 1) : 0u;
            if (t1 != 0u) {
                const unsigned w0 = *(volatile KS_LDS unsigned*)(q + 2 * ticket);     // (written before the task word)
                ticket = 0xffffffffu;
                const int task = (int)t1 - 1, node = (int)(w0 & 0xffffu);
                const int e = task / WG_RAY_TASKS, r = (task % WG_RAY_TASKS) >> RG_BITS, g = 1 + (task & (RG - 1));
                LdsSnap snap{snaps + e * WG_SNAP - SNAP_BP};
                float pnt[3], vec[3], lp[3], lv[3];
                ray_origin(m, snap, r, pnt, vec);
                ray_to_geom(m, snap, g, pnt, vec, lp, lv);
                slot = hit + e * NRAY + r;
                const int mesh = m.geom_mesh[g];
                SharingStack st{LdsStack<float>{stk + tid, NTH}, sh, t1};
                busy = walk.start(m.mesh_tri[mesh], m.mesh_bvh_box[mesh], m.geom_size[g], lp, lv, SlotBound{slot}, st);
                if (node != 0) {                                         // a subtree handed over by another walker: still in front of the nearest hit?
                    const float te = __int_as_float((int)(w0 & 0xffff0000u));
                    const float best = __int_as_float((int)*(volatile KS_LDS unsigned*)slot);
                    busy = busy && te <= best;
                    walk.node = node;
                }
                if (!busy) atomicSub((unsigned*)(ctl + 2), 1u);
            }
        } else {
#ifdef KS_ROLLOUT_STAMP
            my_visits++;
#endif
            if (!walk.step()) {
                if (walk.best >= 0) atomicMin((unsigned*)slot, (unsigned)__float_as_int(walk.best));
                atomicSub((unsigned*)(ctl + 2), 1u);
                busy = false;
            } else if (walk.stack.own.sp > 0 && *(volatile KS_LDS unsigned*)(ctl + 1) > *(volatile KS_LDS unsigned*)ctl) {
                // lanes have run out of work since this walker stacked its subtrees: give one away (push() keeps it if nobody waits any more)
                int pn;
                float pt;
                walk.stack.own.pop(pn, pt);
                walk.stack.push(pn, pt);
            }
        }
        if (__builtin_amdgcn_ballot_w64(busy) == 0ull) __builtin_amdgcn_s_sleep(1);   // a wave with nothing to do: leave the LDS to the others
    }
#ifdef KS_ROLLOUT_STAMP
    if (rprof) {                          // [2] surviving tasks, [3] node visits in total, [4] sum over workgroups of the busiest lane's visits
        if (tid == 0) { atomicAdd((unsigned long long*)&rprof[2], (unsigned long long)nlist); ctl[4] = 0; }
        C::sync();
        atomicAdd((unsigned long long*)&rprof[3], (unsigned long long)my_visits);
        atomicMax((unsigned*)&ctl[4], (unsigned)my_visits);
    }
#endif
    C::sync();
    KS_RP(1)                             // the walks
#ifdef KS_ROLLOUT_STAMP
    if (rprof && tid == 0) { atomicAdd((unsigned long long*)&rprof[4], (unsigned long long)ctl[4]); atomicAdd((unsigned long long*)&rprof[5], (unsigned long long)(ctl[0] - nlist)); }
#endif
    for (int i = tid; i < epw * NRAY; i += NTH) {
        const int e = i / NRAY, r = i % NRAY, env = b.slot_env[slot0 + e];
        if (env >= 0) {
            const float t = __int_as_float((int)hit[i]);
            b.rays[(long)r * N + env] = t < Lim<float>::big ? t : -1.0f;
        }
    }
}

// Where a step's (or reset's) results go
template <typename T> struct ObsOut {
    T* obs; T* reward; uint8_t* done; T* info; T* final_obs;
    int horizon, auto_reset, env_major;
};
template <bool WAVE_SCOPE>
__device__ void wg_obs(const Model<float>& m, const Buffers<float>& b, int N, int slot0, int epw, KS_LDS unsigned* w, const ObsOut<float>& o);
__device__ void wg_ray_pool(const Model<float>* models, const Model<float>& mine, const Buffers<float>& b, int N, int epw, KS_LDS unsigned* w, int n_wg,
                            int linger);
#ifndef KS_POOL_LINGER_LAST
#define KS_POOL_LINGER_LAST 64
#endif
constexpr int POOL_LINGER_LAST = KS_POOL_LINGER_LAST;

// Register budget of the fp32 stepping kernels: 376 of a SIMD lane's 512 (256 VGPRs + 120 AGPRs), so that a wave of the learner's LDS-free kernels
// stays resident BESIDE a stepping wave on every SIMD.  The count the compiler arrives at on its own moves by tens of registers with
// changes far from the hot loops (an out-of-line callee's clobber set decides which of the kernel's live-across-call values sit in AGPRs); at 378 the
// learner no longer fits and training runs at 0.7 x.  amdgpu_num_vgpr(N) caps VGPRs + AGPRs at 2 N on gfx950's unified file; what does not fit is spilled
// to scratch (values that are touched once per substep).
// k_rollout: 376 (beside it: the one-wave learner kernels, 128); k_env_step_f32: 352 (beside it: the lock-step trainer's 4-wave split kernels, 160).
#ifndef KS_ROLLOUT_NUM_VGPR
#define KS_ROLLOUT_NUM_VGPR 188
#endif
#ifndef KS_STEP_NUM_VGPR
#define KS_STEP_NUM_VGPR 176
#endif
#define KS_ROLLOUT_REGS __attribute__((amdgpu_num_vgpr(KS_ROLLOUT_NUM_VGPR)))
#define KS_STEP_REGS __attribute__((amdgpu_num_vgpr(KS_STEP_NUM_VGPR)))
// obs_in_step: the observation / reward / done / auto-reset of the workgroup's envs are produced here too (wg_obs), the
// separate k_obs launch of a step is gone; needs rays_in_step (fp32 / LDS variant).
template <typename T, bool USE_LDS>
__device__ __forceinline__ void env_step_body(const Model<T>* __restrict__ models, const Buffers<T>& b, const Buffers<T>* __restrict__ bdev,
                                              const T* __restrict__ action, int N, int frame_skip, int iters, int epw, int tap, int rays_in_step,
                                              int pair_memory, int obs_in_step, const ObsOut<T>* __restrict__ out, int ray_pool, int n_wg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef KS_STAMP_WG
    const long long wk_entry = wall_clock64();
#endif
    KS_LDS T* lds = (KS_LDS T*)smem;
    // every env of this workgroup holds this object (the id is made wave-uniform by hand: the loads through `mp` are scalar then)
    const Model<T>* mp = models + __builtin_amdgcn_readfirstlane(b.wg_model[blockIdx.x]);
    const Model<T>* ml = mp;
    if constexpr (USE_LDS) { ml = stage_model_and_tables<T, WG>(mp, lds); lds += model_words<T>(); }
    const Model<T>& m = *ml;
    int hull_words = 0;
    // the out-of-line stages reach the descriptor through a generic reference: the workgroup's one copy in LDS (behind the model)
    static_assert(sizeof(Hulls<T>) <= HULLS_BYTES, "Hulls descriptor slot");
    Hulls<T> hu_own;
    Hulls<T>* hup = &hu_own;
    if constexpr (USE_LDS) hup = (Hulls<T>*)(smem + (sizeof(Model<T>) + 15) / 16 * 16);
    // ray pool: how many of the workgroup's 16 team slots have their snapshot out (last 16 bytes of the descriptor slot)
    static_assert(sizeof(Hulls<T>) <= HULLS_BYTES - 16, "Hulls descriptor slot + the arrival counter");
    KS_LDS int* arrive = (KS_LDS int*)(smem + (sizeof(Model<T>) + 15) / 16 * 16 + HULLS_BYTES - 16);
    if (USE_LDS && threadIdx.x == 0) {
        *arrive = 0;
        if (ray_pool) atomicAdd(&b.rayq[3], 1);                     // workgroups of this launch that have started
    }
    if constexpr (USE_LDS) __syncthreads();         // the model copy is complete: the descriptor is built from its counts in LDS
    stage_hulls<T, USE_LDS>(*ml, lds, hull_words, *hup, USE_LDS);
    const Hulls<T>& hu = *hup;
    // epw envs per workgroup, SUBS lanes per env: the lanes of a team keep identical copies of the env state and
    // split the vertex scans / per-pair (collision) and per-contact (solver) loops; per-env dynamic data is shared in LDS
    const int e = threadIdx.x / LANE_STRIDE;
    const Team<SUBS> team{(int)threadIdx.x % LANE_STRIDE};
    const int env = e < epw ? b.slot_env[blockIdx.x * epw + e] : -1;
    const bool active = !(team.sub >= SUBS || env < 0);
    if (!active && !(USE_LDS && rays_in_step)) return;
    // Ray pool (fp32 / LDS, rays in the step): the snapshot the rays need exists once the LAST substep's kinematics are done, a
    // whole substep (~60 us) before the env has finished stepping.  Every team slot reports its snapshot (empty slots at once);
    // the slot that completes the workgroup's 16 publishes the workgroup in the launch's queue, and workgroups that have finished
    // stepping cast the rays of whatever is queued (wg_ray_pool) - so the slowest workgroup's rays are done by others while it
    // computes its last substep, instead of behind it on the critical path.
    auto slot_arrived = [&]() {
        if (!(USE_LDS && ray_pool) || team.sub != 0) return;
        // this team's snapshot stores have reached the L2 (workgroup-scope release: the workgroup's waves share one CU and one L2);
        // the slot that completes the sixteen then makes them visible to the other XCDs' L2s - ONE device-scope release (an L2
        // write-back) per workgroup, not one per team
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (__hip_atomic_fetch_add(arrive, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == EPW_MAX - 1) {
            __threadfence();
            __hip_atomic_store(&b.rayq[4 + n_wg + blockIdx.x], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);     // published
            const int t = atomicAdd(&b.rayq[0], 1);
            __hip_atomic_store(&b.rayq[4 + t], (int)blockIdx.x + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if (!active) slot_arrived();
    if (active) {
    T hq[4], act[4];
    KS_UNROLL
    for (int i = 0; i < 4; i++) { hq[i] = b.hand_quat[(long)i * N + env]; act[i] = action[(long)i * N + env]; }
    int ncon = 0, status = 0;
    ColW<T> snap{b.snap + env, N};
#ifdef KS_STAMP
    float prof[30];
    for (int k = 0; k < 30; k++) prof[k] = 0;
    const long long tk0 = clock64();
    const long long wk0 = wall_clock64();
#else
    float* prof = nullptr;
#endif
    if constexpr (USE_LDS) {
        KS_LDS T* blk = lds + ((hull_words >> 2) << 2) + e * SCR_TOTAL;
        ScratchC<T, KS_LDS T*> scr{blk};
        // state and per-step constants in the env's LDS block, reached through generic pointers
        T* stp = (T*)(blk + SCR_STATE);
        load_state_team<T, SUBS>(b, env, N, stp, team.sub);
        load_env_params(scr, team, b, env, N);
        // Pair memory: what this lane remembers of its hull pairs' narrow-phase queries (GJK simplex, MPR portal: vertex ids)
        // lives in registers across the substeps and comes back from the previous launch, so the first substep of an
        // env-step starts as warm as the other fourteen (a cold GJK is ~10 iterations, a warm one 1-2: the first substep
        // used to cost as much as five of the others).  A stale memory (after a reset) is still a valid start: any ids are.
        constexpr int WPL = (NPAIR_MAX + SUBS - 1) / SUBS;              // pairs per lane
        PairWarm gw[WPL];
        unsigned* pm = b.pairmem + ((long)env * SUBS + team.sub) * (WPL * WARM_WORDS);
        KS_UNROLL
        for (int q = 0; q < WPL; q++) {
            KS_UNROLL
            for (int j = 0; j < WARM_WORDS; j++) gw[q].w[j] = pair_memory ? pm[q * WARM_WORDS + j] : 0u;
        }
        lane_env_step(m, hu, *(LaneState<T>*)stp, hq, act, scr, team, snap, frame_skip, iters, ncon, status, prof, stp + NQ + 2 * NV, gw, slot_arrived);
        if (pair_memory) {
            KS_UNROLL
            for (int q = 0; q < WPL; q++) {
                KS_UNROLL
                for (int j = 0; j < WARM_WORDS; j++) pm[q * WARM_WORDS + j] = gw[q].w[j];
            }
        }
#ifdef KS_STAMP
        // diagnostic build only: per-phase cycle sums of this lane go to the contact tap buffer
        prof[6] = (float)(clock64() - tk0);
#ifdef KS_STAMP_WG
        // wall clock (100 MHz) of the stepping loop's start / end, modulo 2^22 ticks, and the hardware id (CU / SE / XCC)
        prof[13] = (float)(wk0 & 0x3fffff);
        prof[23] = (float)(wk0 - wk_entry);           // table staging + state load, 100 MHz ticks
        prof[14] = (float)(wall_clock64() & 0x3fffff);
        prof[19] = (float)(__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) & 0xffffff);
        prof[20] = (float)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf);
#endif
        if (tap) for (int k = 0; k < 30; k++) b.contact[(long)(k + 30 * team.sub) * N + env] = (T)prof[k];
        tap = 0;
#endif
        if (team.sub == 0)
            for (int k = 0; k < (tap ? ncon * CON_STRIDE : 0); k++) b.contact[(long)k * N + env] = scr(SCR_CON + k);
        team.sync();
        store_state_team<T, SUBS>(b, env, N, stp, team.sub);
    } else {
        LaneState<T> st;
        load_state(b, env, N, st);
        Scratch<T> scr{b.gscratch + env, N};
        load_env_params(scr, team, b, env, N);
        lane_env_step(m, hu, st, hq, act, scr, team, snap, frame_skip, iters, ncon, status);
        if (team.sub == 0) {
            for (int k = 0; k < (tap ? ncon * CON_STRIDE : 0); k++) b.contact[(long)k * N + env] = scr(SCR_CON + k);
            store_state(b, env, N, st);
        }
    }
    if (status) atomicOr(&b.status[env], status);
    if (team.sub == 0) b.ncon[env] = ncon;
    }
    if constexpr (USE_LDS && sizeof(T) == 4) {
        if (rays_in_step) {
            // every thread of the workgroup is here (inactive lanes included): snapshots written, env blocks dead
            __threadfence_block();
            __syncthreads();
            // (the out-of-line tails take the pointer table and the output record by reference: from device memory - a
            // reference to the by-value kernel arguments made every lane copy them to its stack, 224 bytes written through per launch)
            if (ray_pool) wg_ray_pool(models, m, *bdev, N, epw, (KS_LDS unsigned*)(lds + ((hull_words >> 2) << 2)), n_wg, ray_pool == 2);
            else {
                wg_rays<false>(m, *bdev, N, blockIdx.x * epw, epw, (KS_LDS unsigned*)(lds + ((hull_words >> 2) << 2)) KS_RAY_PROF_ARG(nullptr));
                __threadfence_block();
                __syncthreads();
            }
            if (obs_in_step) wg_obs<false>(m, *bdev, N, blockIdx.x * epw, epw, (KS_LDS unsigned*)(lds + ((hull_words >> 2) << 2)), *out);
            if (ray_pool) {
                // the last workgroup to leave clears the pool for the next launch
                KS_LDS int* last = (KS_LDS int*)(lds + ((hull_words >> 2) << 2));
                __syncthreads();
                if (threadIdx.x == 0) *last = atomicAdd(&b.rayq[2], 1) == n_wg - 1;
                __syncthreads();
                if (*last) {
                    for (int i = threadIdx.x; i < 2 * n_wg; i += WG) b.rayq[4 + i] = 0;
                    if (threadIdx.x < 4) b.rayq[threadIdx.x] = 0;
                }
            }
        }
    }
}


// the kernel: the fp64 instantiation takes what it needs (256 + 256 registers, the parity instrument); the fp32 product is capped (KS_STEP_NUM_VGPR above)
template <typename T, bool USE_LDS>
__global__ __launch_bounds__(WG) void k_env_step(const Model<T>* __restrict__ models, Buffers<T> b, const Buffers<T>* __restrict__ bdev,
                                                   const T* __restrict__ action, int N, int frame_skip, int iters, int epw, int tap, int rays_in_step,
                                                   int pair_memory, int obs_in_step, const ObsOut<T>* __restrict__ out, int ray_pool, int n_wg) {
    env_step_body<T, USE_LDS>(models, b, bdev, action, N, frame_skip, iters, epw, tap, rays_in_step, pair_memory, obs_in_step, out, ray_pool, n_wg);
}
__global__ __launch_bounds__(WG) KS_STEP_REGS void k_env_step_f32(const Model<float>* __restrict__ models, Buffers<float> b, const Buffers<float>* __restrict__ bdev,
                                                                       const float* __restrict__ action, int N, int frame_skip, int iters, int epw, int tap, int rays_in_step,
                                                                       int pair_memory, int obs_in_step, const ObsOut<float>* __restrict__ out, int ray_pool, int n_wg) {
    env_step_body<float, true>(models, b, bdev, action, N, frame_skip, iters, epw, tap, rays_in_step, pair_memory, obs_in_step, out, ray_pool, n_wg);
}


// ---- free-running rollout (include/kinova_sim.h: ks_rollout): k_env_step's fp32 / LDS path inside a per-workgroup loop, with
// the actor forward + action selection in front of the step and the replay write behind it.  No cross-workgroup state: the ray
// pool is off (every workgroup casts its own envs' rays).  The two added phases are out-of-line and reach their arguments
// through a pointer to device memory: nothing of them stays in registers across the stepping phase, whose footprint (344 of
// the 512 registers per lane) is what lets the learner's LDS-free waves run beside this kernel for its whole life.
template <int NT1, int NT2>
__device__ __noinline__ void rollout_policy(const ks_rollout_args* __restrict__ rap, int N, int row_env, KS_LDS float* blocks) {
    const ks_rollout_args& ra = *rap;
    const int S = krsel::S, A = krsel::A;
    // the policy's scratch lives in the (then dead) env blocks: H1 [NT1 * 4][16], H2 [NT2 * 4][16], P [4][16] float4
    kmlp::f32x4(*H1)[kmlp::ROWS] = (kmlp::f32x4(*)[kmlp::ROWS])(float*)blocks;
    kmlp::f32x4(*H2)[kmlp::ROWS] = H1 + NT1 * 4;
    kmlp::f32x4(*Pp)[kmlp::ROWS] = H2 + NT2 * 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // one word pair behind the scratch: the version this workgroup uses, and whether it went stale - decided by ONE thread and
    // read by all after a barrier, so that the four waves can neither mix weight buffers nor disagree on repeating the forward
    long long* vbox = (long long*)(Pp + kmlp::NW);          // (not volatile: that costs 28 registers here; the barriers order the accesses)
    for (;;) {
        if (threadIdx.x == 0) vbox[0] = (long long)__hip_atomic_load(ra.actor_ver, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const long long ver = vbox[0];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                      // the newest complete weight buffer, as written
        const float* pw = ra.actor_pub + (ver % 3) * ra.actor_stride;
        kmlp::f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        const bool mine = kmlp::mlp3_rows16<NT1, NT2, true>(wave, lane, (long)row_env, S, 0, ra.h1, ra.h2, A, ra.obs, S, nullptr, 0, pw + ra.off_w1,
                                                            pw + ra.off_b1, pw + ra.off_w2, pw + ra.off_b2, pw + ra.off_w3, nullptr, nullptr, H1, H2, Pp,
                                                            z4);
        // the last words read from buffer ver % 3 - the layer-3 bias - are taken BEFORE the staleness check below, and a fence keeps
        // the check's load behind them: everything the action is computed from has then been read when the counter is looked at
        float y[4] = {0.f, 0.f, 0.f, 0.f};
        if (mine) {
            const float z[4] = {z4.x, z4.y, z4.z, z4.w};
            const float* b3 = pw + ra.off_b3;
#pragma unroll
            for (int i = 0; i < 4; i++) y[i] = ra.max_action / (1.f + __expf(-(z[i] + b3[i])));
        }
        // ALL FOUR waves' reads of buffer ver % 3 must be complete before the counter is looked at again: the barrier waits for every
        // wave to reach this point (each wave's loads are consumed by the arithmetic above, so they have returned), and the agent-scope
        // acquire fence keeps thread 0's second load of the counter - written by another kernel - behind it.
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // buffer ver % 3 is rewritten by publish ver + 3, which starts once ver + 2 is complete: if the counter has advanced by two
        // while this forward ran, the weights just read may be torn - repeat with the newest ones (two update periods, > 1 ms,
        // against a 25 us forward: never seen; the check makes it a protocol instead of a timing assumption)
        if (threadIdx.x == 0) vbox[1] = (long long)__hip_atomic_load(ra.actor_ver, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ver;
        __syncthreads();
        const bool stale = vbox[1] >= 2;
        if (!stale) {
            if (mine) {
                float nz[4];
                krsel::normal4(ra.seed, (unsigned long long)ra.steps_total[row_env], (uint32_t)row_env, nz);
                krsel::select_one(row_env, N, y, nz, ra.obs, ra.prev_obs, ra.has_prev, ra.t, ra.ready, ra.sigma, ra.max_action, ra.skip_steps, ra.action,
                                  ra.action_t, ra.lifting);
            }
            return;
        }
        __syncthreads();
    }
}

// replay write + per-env bookkeeping of one env by its 16-lane team (k_store_transition of ks_rollout.hip + the episode hand-over)
__device__ __noinline__ void rollout_store(const ks_rollout_args* __restrict__ rap, int N, int i, int sub) {
    const ks_rollout_args& ra = *rap;
    const int S = krsel::S, A = krsel::A, H = ra.horizon;
    const bool done = ra.sim_done[i] != 0, lift = ra.lifting[i] != 0;
    const float rew = ra.sim_reward[i];
    const bool store = ra.with_replay && !lift;
    const int sel = ra.with_replay ? ra.cur_sel[i] : 0;
    const long bi = (long)sel * N + i;
    const long len0 = ra.with_replay ? ra.cur_len[bi] : 0;
    const long tt = len0 < H - 1 ? len0 : H - 1;
    const long row = bi * H + tt;
    for (int c = sub; c < S; c += SUBS) {
        const float so = ra.sim_obs[(long)i * S + c];
        const float st = ra.obs[(long)i * S + c];
        const float nx = done ? ra.sim_final_obs[(long)i * S + c] : so;          // (auto-reset contexts: the terminal observation)
        if (store) { ra.cur_state[row * S + c] = st; ra.cur_next[row * S + c] = nx; }
        ra.prev_obs[(long)i * S + c] = done ? so : st;
        ra.obs[(long)i * S + c] = so;
    }
    if (store && sub < A) ra.cur_action[row * A + sub] = ra.action[(long)i * A + sub];
    if (sub != 0) return;
    long len1 = len0;
    if (store) {
        ra.cur_reward[row] = rew;
        ra.cur_not_done[row] = done ? 0.0f : 1.0f;
        len1 = tt + 1;
    }
    if (ra.with_replay) {
        if (done && lift && len1 > 0) {                                         // ended during the scripted lift: the outcome goes
            ra.cur_reward[bi * H + len1 - 1] = rew;                              // into the last stored transition (utils.py:309-343)
            ra.cur_not_done[bi * H + len1 - 1] = 0.0f;
        }
        ra.cur_len[bi] = len1;
        if (done) {
            const bool keep = len1 - ra.n_steps > 1;
            const long bo = (long)(sel ^ 1) * N + i;
            const bool free_other = __hip_atomic_load(&ra.pub_len[bo], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0;
#ifdef KS_DEBUG_DROPS
            // diagnostic build: counters[] has N more words behind its 8 + 4 * 512 + 8; [4] sum of env-steps since the env's last publication at a
            // drop, [5] drops, [6] smallest such gap, [7] drops whose other buffer reads as free on a second, later look
            int64_t* dbg = ra.counters + 8 + 4 * 512 + 8;
            if (keep && !free_other) {
                const long long gap = (long long)ra.steps_total[i] - (long long)dbg[i];
                atomicAdd((unsigned long long*)&ra.counters[4], (unsigned long long)gap);
                atomicAdd((unsigned long long*)&ra.counters[5], 1ull);
                atomicMin((long long*)&ra.counters[6], gap);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (__hip_atomic_load(&ra.pub_len[bo], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicAdd((unsigned long long*)&ra.counters[7], 1ull);
            }
            if (keep && free_other) dbg[i] = ra.steps_total[i];
#endif
            if (keep && free_other) {
                __threadfence();                                                // (the team's row stores were fenced by the caller)
                __hip_atomic_store(&ra.pub_len[bi], (int64_t)len1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                ra.cur_sel[i] = (uint8_t)(sel ^ 1);
                ra.cur_len[bo] = 0;
                atomicAdd((unsigned long long*)&ra.counters[2], 1ull);
            } else {
                ra.cur_len[bi] = 0;
                if (keep) atomicAdd((unsigned long long*)&ra.counters[3], 1ull);
            }
        }
    }
    if (done) {
        atomicAdd((unsigned long long*)&ra.counters[0], 1ull);
        if (ra.sim_done[i] & 1) atomicAdd((unsigned long long*)&ra.counters[1], 1ull);
    }
    ra.has_prev[i] = !done;
    ra.t[i] = done ? 0 : ra.t[i] + 1;
    __hip_atomic_store(&ra.steps_total[i], ra.steps_total[i] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (device-visible: kr_wait_min polls it from another stream)
    ra.ready[i] = (ra.ready[i] != 0) && !done;
    ra.reward_out[i] = rew;
    ra.done_out[i] = done;
}

// k_rollout's ready queue (more groups than resident workgroups): [0] head ticket, [1] tail ticket, [ROLLOUT_Q_STEPS + g] env-steps group g has done in
// this launch, [ROLLOUT_Q_RING ..] the ring of lap-tagged group ids
constexpr int ROLLOUT_QCAP = 2048, ROLLOUT_Q_STEPS = 16, ROLLOUT_Q_RING = ROLLOUT_Q_STEPS + ROLLOUT_QCAP, ROLLOUT_Q_WORDS = ROLLOUT_Q_RING + ROLLOUT_QCAP;
__global__ void k_rollout_queue_init(int* __restrict__ queue, int n_groups) {
    for (int i = threadIdx.x; i < ROLLOUT_Q_WORDS; i += blockDim.x) {
        int v = 0;
        if (i == 1) v = n_groups;                                                                  // tail: the first push goes behind the initial fill
        if (i >= ROLLOUT_Q_RING && i - ROLLOUT_Q_RING < n_groups) v = (1 << 12) | (i - ROLLOUT_Q_RING + 1);     // lap 1, group i
        queue[i] = v;
    }
}

// One env-step of the workgroup's envs (the body of k_rollout's loop).
template <int NT1, int NT2>
__device__ __forceinline__ void rollout_iter(const Model<float>& m, const Hulls<float>& hu, const Buffers<float>* __restrict__ bdev, int N, int frame_skip,
                                          int iters, int epw, int pair_memory, const ObsOut<float>* __restrict__ out,
                                          const ks_rollout_args* __restrict__ rap, KS_LDS float* blocks, int grp) {
    using T = float;
    const Buffers<T>& b = *bdev;
    const int e = threadIdx.x / LANE_STRIDE;
    const Team<SUBS> team{(int)threadIdx.x % LANE_STRIDE};
    // (the env ids are re-read from the slot list in every iteration rather than kept in registers across the stepping phase)
    // grp: the 16-env group of the slot list this workgroup steps now (its own, blockIdx.x, unless the launch has fewer workgroups than groups)
    const int env = e < epw ? b.slot_env[grp * epw + e] : -1;
    const bool active = !(team.sub >= SUBS || env < 0);
    KS_LDS unsigned* w = (KS_LDS unsigned*)blocks;
#ifdef KS_ROLLOUT_STAMP
    // diagnostic build: wall-clock ticks (100 MHz) of the four phases, summed over workgroups and env-steps in counters[4..7]
    long long tk = wall_clock64();
#define KS_RS(i) { const long long t1_ = wall_clock64(); if (threadIdx.x == 0) atomicAdd((unsigned long long*)&rap->counters[4 + i], (unsigned long long)(t1_ - tk)); tk = t1_; }
#else
#define KS_RS(i)
#endif
    {
        const int nn = threadIdx.x & 15;
        const int row_env = nn < epw ? b.slot_env[grp * epw + nn] : -1;   // the policy row of this lane (the same in all four waves)
        rollout_policy<NT1, NT2>(rap, N, row_env, blocks);
    }
    __threadfence_block();
    __syncthreads();
    KS_RS(0)
    // ---- the env-step (k_env_step's LDS path)
    if (active) {
        T hq[4], act[4];
        KS_UNROLL
        for (int i = 0; i < 4; i++) { hq[i] = b.hand_quat[(long)i * N + env]; act[i] = rap->action_t[(long)i * N + env]; }
        int ncon = 0, status = 0;
        ColW<T> snap{b.snap + env, N};
        KS_LDS T* blk = blocks + e * SCR_TOTAL;
        ScratchC<T, KS_LDS T*> scr{blk};
        T* stp = (T*)(blk + SCR_STATE);
        load_state_team<T, SUBS>(b, env, N, stp, team.sub);
        load_env_params(scr, team, b, env, N);
        constexpr int WPL = (NPAIR_MAX + SUBS - 1) / SUBS;
        PairWarm gw[WPL];
        unsigned* pm = b.pairmem + ((long)env * SUBS + team.sub) * (WPL * WARM_WORDS);
        KS_UNROLL
        for (int q = 0; q < WPL; q++) {
            KS_UNROLL
            for (int j = 0; j < WARM_WORDS; j++) gw[q].w[j] = pair_memory ? pm[q * WARM_WORDS + j] : 0u;
        }
        lane_env_step(m, hu, *(LaneState<T>*)stp, hq, act, scr, team, snap, frame_skip, iters, ncon, status, (float*)nullptr, stp + NQ + 2 * NV, gw,
                      []() {});
        if (pair_memory) {
            KS_UNROLL
            for (int q = 0; q < WPL; q++) {
                KS_UNROLL
                for (int j = 0; j < WARM_WORDS; j++) pm[q * WARM_WORDS + j] = gw[q].w[j];
            }
        }
        team.sync();
        store_state_team<T, SUBS>(b, env, N, stp, team.sub);
        if (status) atomicOr(&b.status[env], status);
        if (team.sub == 0) b.ncon[env] = ncon;
    }
    __threadfence_block();
    __syncthreads();
    KS_RS(1)
    wg_rays<false>(m, b, N, grp * epw, epw, w KS_RAY_PROF_ARG((long long*)&rap->counters[8 + 4 * 512]));
    __threadfence_block();
    __syncthreads();
    KS_RS(2)
    wg_obs<false>(m, b, N, grp * epw, epw, w, *out);
    __threadfence_block();
    __syncthreads();
    if (active) rollout_store(rap, N, env, team.sub);
    __threadfence_block();
    __syncthreads();
    KS_RS(3)
#undef KS_RS
}

// ---- round 6: the waves of a workgroup run free.  rollout_iter above joins the four waves of a workgroup five times per env-step
// (policy | 15 substeps | rays | observation | replay write), so every env-step costs the workgroup its SLOWEST wave (measured model,
// profiles/r05_wave_sorting.txt: workgroup cost 1408 k cycles per env-step against 1091 k for the mean wave).  Nothing in an env-step
// needs the other waves: per-env data is private to the wave's own four env blocks, the model and the hull tables are read-only.
// Here the unit of the persistent loop is ONE WAVE with its four envs: the actor forward for its own four rows (a 16-column MFMA
// tile a quarter full, bit-equal per row to the 4-wave tile), its own rays (the same work-sharing queue among 64 lanes), its own
// observation and replay rows - and no s_barrier between kernel entry and exit.  Scratch of the tails: the wave's own four (then
// dead) env blocks.
__device__ __forceinline__ long long uniform64(long long v) {      // lane 0's value, in scalar registers
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
template <int NT1, int NT2>
__device__ __noinline__ void rollout_policy_wave(const ks_rollout_args* __restrict__ rap, int N, int row_env, KS_LDS float* wblocks) {
    const ks_rollout_args& ra = *rap;
    const int S = krsel::S, A = krsel::A;
    constexpr int NR = 4;
    kmlp::f32x4(*H1)[NR] = (kmlp::f32x4(*)[NR])(float*)wblocks;
    kmlp::f32x4(*H2)[NR] = H1 + NT1 * 4;
    const int lane = threadIdx.x & 63;
    for (;;) {
        // (the wave's own look at the version counter: waves of a workgroup may act on different - each complete - weight versions)
        const long long ver = uniform64(__hip_atomic_load(ra.actor_ver, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const float* pw = ra.actor_pub + (ver % 3) * ra.actor_stride;
        kmlp::f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        const bool mine = kmlp::mlp3_rows_wave<NT1, NT2, true, NR>(lane, (long)row_env, S, ra.h1, ra.h2, A, ra.obs, S, pw + ra.off_w1, pw + ra.off_b1,
                                                                     pw + ra.off_w2, pw + ra.off_b2, pw + ra.off_w3, H1, H2, z4);
        float y[4] = {0.f, 0.f, 0.f, 0.f};
        if (mine) {
            const float z[4] = {z4.x, z4.y, z4.z, z4.w};
            const float* b3 = pw + ra.off_b3;
#pragma unroll
            for (int i = 0; i < 4; i++) y[i] = ra.max_action / (1.f + __expf(-(z[i] + b3[i])));
        }
        // every read of buffer ver % 3 by this wave has been consumed by the arithmetic above; the acquire keeps the second look at
        // the counter behind them (see rollout_policy: a buffer is rewritten two publications later)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const long long ver2 = uniform64(__hip_atomic_load(ra.actor_ver, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (ver2 - ver < 2) {
            if (mine) {
                float nz[4];
                krsel::normal4(ra.seed, (unsigned long long)ra.steps_total[row_env], (uint32_t)row_env, nz);
                krsel::select_one(row_env, N, y, nz, ra.obs, ra.prev_obs, ra.has_prev, ra.t, ra.ready, ra.sigma, ra.max_action, ra.skip_steps, ra.action,
                                  ra.action_t, ra.lifting);
            }
            return;
        }
    }
}

// One env-step of ONE WAVE's four envs (slots grp * epw + 4 * wave ..): the body of k_rollout's barrier-free loop.
template <int NT1, int NT2>
__device__ __forceinline__ void rollout_iter_wave(const Model<float>& m, const Hulls<float>& hu, const Buffers<float>* __restrict__ bdev, int N, int frame_skip,
                                               int iters, int epw, int pair_memory, const ObsOut<float>* __restrict__ out,
                                               const ks_rollout_args* __restrict__ rap, KS_LDS float* blocks, int grp) {
    using T = float;
    using C = Crew<true>;
    const Buffers<T>& b = *bdev;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int e = threadIdx.x / LANE_STRIDE;
    const Team<SUBS> team{(int)threadIdx.x % LANE_STRIDE};
    const int env = e < epw ? b.slot_env[grp * epw + e] : -1;
    const bool active = env >= 0;
    const int wslot0 = grp * epw + 4 * wave, wepw = epw - 4 * wave < 4 ? (epw - 4 * wave < 0 ? 0 : epw - 4 * wave) : 4;   // this wave's slots
    KS_LDS T* wblocks = blocks + 4 * wave * SCR_TOTAL;
    KS_LDS unsigned* w = (KS_LDS unsigned*)wblocks;
    {
        const int nn = lane & 15;
        const int row_env = nn < wepw ? b.slot_env[wslot0 + nn] : -1;
        rollout_policy_wave<NT1, NT2>(rap, N, row_env, wblocks);
    }
    C::sync();
    if (active) {
        T hq[4], act[4];
        KS_UNROLL
        for (int i = 0; i < 4; i++) { hq[i] = b.hand_quat[(long)i * N + env]; act[i] = rap->action_t[(long)i * N + env]; }
        int ncon = 0, status = 0;
        ColW<T> snap{b.snap + env, N};
        KS_LDS T* blk = blocks + e * SCR_TOTAL;
        ScratchC<T, KS_LDS T*> scr{blk};
        T* stp = (T*)(blk + SCR_STATE);
        load_state_team<T, SUBS>(b, env, N, stp, team.sub);
        load_env_params(scr, team, b, env, N);
        constexpr int WPL = (NPAIR_MAX + SUBS - 1) / SUBS;
        PairWarm gw[WPL];
        unsigned* pm = b.pairmem + ((long)env * SUBS + team.sub) * (WPL * WARM_WORDS);
        KS_UNROLL
        for (int q = 0; q < WPL; q++) {
            KS_UNROLL
            for (int j = 0; j < WARM_WORDS; j++) gw[q].w[j] = pair_memory ? pm[q * WARM_WORDS + j] : 0u;
        }
        lane_env_step(m, hu, *(LaneState<T>*)stp, hq, act, scr, team, snap, frame_skip, iters, ncon, status, (float*)nullptr, stp + NQ + 2 * NV, gw,
                      []() {});
        if (pair_memory) {
            KS_UNROLL
            for (int q = 0; q < WPL; q++) {
                KS_UNROLL
                for (int j = 0; j < WARM_WORDS; j++) pm[q * WARM_WORDS + j] = gw[q].w[j];
            }
        }
        team.sync();
        store_state_team<T, SUBS>(b, env, N, stp, team.sub);
        if (status) atomicOr(&b.status[env], status);
        if (team.sub == 0) b.ncon[env] = ncon;
    }
    C::sync();
    wg_rays<true>(m, b, N, wslot0, wepw, w KS_RAY_PROF_ARG(nullptr));
    C::sync();
    wg_obs<true>(m, b, N, wslot0, wepw, w, *out);
    C::sync();
    if (active) rollout_store(rap, N, env, team.sub);
    C::sync();
}

// another object's model constants and hull tables into the workgroup's LDS (out of line: nothing of it may stay in registers across
// the stepping phase); returns the offset of the env blocks behind the tables, in words
__device__ __noinline__ int rollout_restage(const Model<float>* __restrict__ mp, KS_LDS float* lds0, Hulls<float>* hup) {
    __syncthreads();
    const Model<float>* ml = stage_model_and_tables<float, WG>(mp, lds0);
    int hull_words = 0;
    __syncthreads();
    stage_hulls<float, true>(*ml, lds0 + model_words<float>(), hull_words, *hup, true);
    return (hull_words >> 2) << 2;
}

template <int NT1, int NT2>
__global__ __launch_bounds__(WG) KS_ROLLOUT_REGS void k_rollout(const Model<float>* __restrict__ models, Buffers<float> b, const Buffers<float>* __restrict__ bdev, int N,
                                                int frame_skip, int iters, int epw, int pair_memory, const ObsOut<float>* __restrict__ out,
                                                const ks_rollout_args* __restrict__ rap, int n_iter, int n_groups, int* __restrict__ queue, int wave_free) {
    using T = float;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef KS_ROLLOUT_STAMP
    const long long wk_entry = wall_clock64();      // diagnostic build: counters[] is [8 + 4 * 512 + 8] there (pipeline.AsyncTrainer allocates that)
#endif
    // The 16-env groups of the slot list [n_groups] are dealt to the launch's workgroups in contiguous runs (the list is sorted by
    // object, so a workgroup's groups mostly share their hull tables): one group per workgroup when they all fit the GPU at once
    // (4096 envs on 256 CUs), two or three when they do not (BASELINE config 5: 8192 envs) - a persistent workgroup then takes its
    // groups' env-steps in turn, every group always on the same CU (its state never changes caches), instead of a second round of
    // workgroups that could only start when a first-round workgroup had finished ALL its env-steps.
    // (n_groups < 0: the groups are dealt round-robin instead - workgroup w steps groups w, w + G, w + 2 G ... - which mixes objects
    // within a workgroup: more table restaging, but cheap and expensive objects average out over a workgroup's groups)
    const bool deal_rr = n_groups < 0;
    if (deal_rr) n_groups = -n_groups;
    const int g0 = deal_rr ? (int)blockIdx.x : (int)((long)blockIdx.x * n_groups / gridDim.x);
    const int g1 = deal_rr ? n_groups : (int)((long)(blockIdx.x + 1) * n_groups / gridDim.x);
    const int gstep = deal_rr ? (int)gridDim.x : 1;
    KS_LDS T* const lds0 = (KS_LDS T*)smem;
    KS_LDS T* const lds = lds0 + model_words<T>();
    Hulls<T>* const hup = (Hulls<T>*)(smem + (sizeof(Model<T>) + 15) / 16 * 16);
    int staged = __builtin_amdgcn_readfirstlane(b.wg_model[g0]);
    const Model<T>* ml = stage_model_and_tables<T, WG>(models + staged, lds0);
    int hull_words = 0;
    __syncthreads();
    stage_hulls<T, true>(*ml, lds, hull_words, *hup, true);
    KS_LDS T* blocks = lds + ((hull_words >> 2) << 2);                 // the 16 env blocks; scratch of the tails between the steps
#ifdef KS_ROLLOUT_STAMP
    const long long wk_loop = wall_clock64();
    const long long ck_loop = clock64();
#endif
    if (wave_free) {
        // ---- ONE GROUP PER WORKGROUP (4096 envs on 256 CUs, the metric's shape), round 6: from here on the four waves never meet again -
        // each loops over its own four envs (rollout_iter_wave) and leaves when it has done its n_iter env-steps.
        // (opt-in time budget, ks_rollout_args.budget_ticks: the wave starts no further env-step once the budget has passed - at least one)
        const long long budget = rap->budget_ticks, t_end = wall_clock64() + budget;
#pragma clang loop unroll(disable)
        for (int it = 0; it < n_iter; it++) {
            if (budget > 0 && it > 0 && wall_clock64() - t_end > 0) break;
            const Model<T>* mi = ml;
            Hulls<T>* hi = hup;
            const Buffers<T>* bi = bdev;
            const ObsOut<T>* oi = out;
            const ks_rollout_args* ri = rap;
            KS_LDS T* ki = blocks;
            int gi = g0;
            asm volatile("" : "+s"(mi), "+s"(hi), "+s"(bi), "+s"(oi), "+s"(ri), "+s"(gi));
            asm volatile("" : "+v"(ki));
            rollout_iter_wave<NT1, NT2>(*mi, *hi, bi, N, frame_skip, iters, epw, pair_memory, oi, ri, ki, gi);
        }
        return;
    }
#ifdef KS_ROLLOUT_WAVES_ONLY
    // (A/B build, tools/r06: only the free-wave form is compiled in - ONE inlined copy of the env-step body instead of three - to measure what the
    // scalar-register spills of the three-copy kernel cost; such a build cannot run contexts with more groups than workgroups)
    return;
#else
    if (queue != nullptr) {
        // ---- MORE GROUPS THAN RESIDENT WORKGROUPS, round 5: a FIFO of READY groups instead of a fixed deal.  The ring starts with every group
        // (k_rollout_queue_init); a workgroup pops the group at the head, steps it ONCE, and - unless that was the group's last env-step of the
        // launch - pushes it back at the tail.  A popped group is ready by construction (nobody waits for a predecessor), groups take turns in
        // completion order (every env still does exactly n_iter env-steps per launch), and a workgroup is never idle while a group is ready: a
        // launch costs sum(group-steps) / workgroups instead of (groups per workgroup) x the slowest group.  With the fixed deal 526 groups on 256
        // CUs (BASELINE config 5 drawn per env: 14 partly filled groups) paced the launch at THREE group-steps per env-step, and a stage's bowls
        // (4.5 ms per env-step against 1.3 ms for its cubes) paced every workgroup that held one.
        // The group's state travels through global memory as before (rollout_iter loads it at the start and stores it at the end of every
        // env-step); what is new is that the next env-step may run on another CU, possibly of another XCD: the workgroup that finishes a group
        // releases at agent scope (every wave: its own stores; buffer_wbl2) before the push, the one that pops it acquires before its first load.
        // Ring entries carry their lap ((ticket / QCAP + 1) << 12 | group + 1): a popper spins until ITS lap's entry is there - only when fewer
        // groups are ready than workgroups free, i.e. when it would idle anyway.
        __shared__ int s_grp;
        int* const q_head = queue, * const q_tail = queue + 1, * const q_steps = queue + ROLLOUT_Q_STEPS;
        unsigned* const ring = (unsigned*)(queue + ROLLOUT_Q_RING);
        const int total = n_groups * n_iter;
#pragma clang loop unroll(disable)
        for (;;) {
            __syncthreads();
            if (threadIdx.x == 0) {
                const int t = atomicAdd(q_head, 1);
                int grp = -1;
                if (t < total) {
                    const unsigned lap = (unsigned)(t / ROLLOUT_QCAP) + 1u;
                    unsigned v = __hip_atomic_load(&ring[t % ROLLOUT_QCAP], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    while ((v >> 12) != lap) {
                        __builtin_amdgcn_s_sleep(16);
                        v = __hip_atomic_load(&ring[t % ROLLOUT_QCAP], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    grp = (int)(v & 4095u) - 1;
                }
                s_grp = grp;
            }
            __syncthreads();
            const int grp = __builtin_amdgcn_readfirstlane(s_grp);
            if (grp < 0) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");              // the group's state as its last env-step - on whatever CU - left it
            const int want = __builtin_amdgcn_readfirstlane(bdev->wg_model[grp]);
            if (want != staged) {
                staged = want;
                blocks = lds + rollout_restage(models + staged, lds0, hup);
            }
            const Model<T>* mi = ml;
            Hulls<T>* hi = hup;
            const Buffers<T>* bi = bdev;
            const ObsOut<T>* oi = out;
            const ks_rollout_args* ri = rap;
            KS_LDS T* ki = blocks;
            int gi = grp;
            asm volatile("" : "+s"(mi), "+s"(hi), "+s"(bi), "+s"(oi), "+s"(ri), "+s"(gi));
            asm volatile("" : "+v"(ki));
            rollout_iter<NT1, NT2>(*mi, *hi, bi, N, frame_skip, iters, epw, pair_memory, oi, ri, ki, gi);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");              // every wave: its stores of this env-step, before the group is handed on
            __syncthreads();
            if (threadIdx.x == 0) {
                const int c = q_steps[gi] + 1;                               // (only the group's current owner touches its counter)
                q_steps[gi] = c;
                if (c < n_iter) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    const int t2 = atomicAdd(q_tail, 1);
                    __hip_atomic_store(&ring[t2 % ROLLOUT_QCAP], (((unsigned)(t2 / ROLLOUT_QCAP) + 1u) << 12) | (unsigned)(gi + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        return;
    }
    // The loop's pointers are laundered through empty asm statements at the top of every iteration: otherwise the compiler hoists
    // loop-invariant address arithmetic and model constants out of the loop and keeps them in registers for the whole launch
    // (388 registers per lane instead of ~344; as an out-of-line call the body saves its callee-saved registers to scratch:
    // 1.5 KB per lane) - and the learner's waves (<= 168 registers) must fit beside this kernel on every SIMD.
#pragma clang loop unroll(disable)
    for (int it = 0; it < n_iter; it++) {
#pragma clang loop unroll(disable)
        for (int grp = g0; grp < g1; grp += gstep) {
            if (g1 - g0 > gstep) {
                // another group of this workgroup: restage the tables when its object differs from the one in LDS (~10 us)
                const int want = __builtin_amdgcn_readfirstlane(bdev->wg_model[grp]);
                if (want != staged) {
                    staged = want;
                    blocks = lds + rollout_restage(models + staged, lds0, hup);
                }
            }
            const Model<T>* mi = ml;
            Hulls<T>* hi = hup;
            const Buffers<T>* bi = bdev;
            const ObsOut<T>* oi = out;
            const ks_rollout_args* ri = rap;
            KS_LDS T* ki = blocks;
            int gi = grp;
            asm volatile("" : "+s"(mi), "+s"(hi), "+s"(bi), "+s"(oi), "+s"(ri), "+s"(gi));
            asm volatile("" : "+v"(ki));
            rollout_iter<NT1, NT2>(*mi, *hi, bi, N, frame_skip, iters, epw, pair_memory, oi, ri, ki, gi);
        }
    }
#ifdef KS_ROLLOUT_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 512) {
        // per workgroup, LAST launch: entry (absolute ticks), entry -> loop, loop duration; [8 + 1536 ..]: first entry / last exit of the launch
        const long long wk_end = wall_clock64();
        rap->counters[8 + blockIdx.x] = wk_entry;
        rap->counters[8 + 512 + blockIdx.x] = wk_loop - wk_entry;
        rap->counters[8 + 1024 + blockIdx.x] = wk_end - wk_loop;
        rap->counters[8 + 1536 + blockIdx.x] = clock64() - ck_loop;        // shader-clock cycles of the loop: / its wall time = the clock the CU ran at
    }
#endif
#endif
}

template <typename T, bool USE_LDS>
__global__ __launch_bounds__(WG) void k_substep(const Model<T>* __restrict__ models, Buffers<T> b, const T* __restrict__ ctrl, int N, int iters, int epw, int tap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    KS_LDS T* lds = (KS_LDS T*)smem;
    const Model<T>* mp = models + b.wg_model[blockIdx.x];
    const Model<T>* ml = mp;
    if constexpr (USE_LDS) { ml = stage_model(mp, lds); lds += model_words<T>(); }
    const Model<T>& m = *ml;
    int hull_words = 0;
    // the out-of-line stages reach the descriptor through a generic reference: the workgroup's one copy in LDS (behind the model)
    static_assert(sizeof(Hulls<T>) <= HULLS_BYTES, "Hulls descriptor slot");
    Hulls<T> hu_own;
    Hulls<T>* hup = &hu_own;
    if constexpr (USE_LDS) hup = (Hulls<T>*)(smem + (sizeof(Model<T>) + 15) / 16 * 16);
    stage_hulls<T, USE_LDS>(*mp, lds, hull_words, *hup);
    const Hulls<T>& hu = *hup;
    const int e = threadIdx.x / LANE_STRIDE;
    const Team<SUBS> team{(int)threadIdx.x % LANE_STRIDE};
    const int env = e < epw ? b.slot_env[blockIdx.x * epw + e] : -1;
    if (team.sub >= SUBS || env < 0) return;
    LaneState<T> st;
    load_state(b, env, N, st);
    T hq[4], c[NU], R7[9];
    KS_UNROLL
    for (int i = 0; i < 4; i++) hq[i] = b.hand_quat[(long)i * N + env];
    KS_UNROLL
    for (int i = 0; i < NU; i++) c[i] = ctrl[(long)i * N + env];
    hand_rotation(hq, R7);
    int ncon = 0, status = 0;
    if constexpr (USE_LDS) {
        ScratchC<T, KS_LDS T*> scr{lds + ((hull_words >> 2) << 2) + e * SCR_TOTAL};
        load_env_params(scr, team, b, env, N);
        reset_pair_words<T>(scr, team);
        mj_forward_step(m, hu, st.qpos, st.qvel, st.warm, c, R7, scr, team, iters, true, ncon, status);
        if (team.sub == 0)
            for (int k = 0; k < (tap ? ncon * CON_STRIDE : 0); k++) b.contact[(long)k * N + env] = scr(SCR_CON + k);
    } else {
        Scratch<T> scr{b.gscratch + env, N};
        load_env_params(scr, team, b, env, N);
        reset_pair_words<T>(scr, team);
        mj_forward_step(m, hu, st.qpos, st.qvel, st.warm, c, R7, scr, team, iters, true, ncon, status);
        if (team.sub == 0)
            for (int k = 0; k < (tap ? ncon * CON_STRIDE : 0); k++) b.contact[(long)k * N + env] = scr(SCR_CON + k);
    }
    if (status) atomicOr(&b.status[env], status);
    if (team.sub != 0) return;
    store_state(b, env, N, st);
    b.ncon[env] = ncon;
}

// (re)initialise flagged envs from their stored initial state
template <typename T, bool USE_LDS>
__global__ __launch_bounds__(WAVE) void k_reset(const Model<T>* __restrict__ models, Buffers<T> b, int N, int clear_flag) {
    __shared__ T lds[SCR_CON * WAVE];       // forward kinematics only touches the body-pose part of the scratch
    const int env = blockIdx.x * WAVE + threadIdx.x;
    if (env >= N || !b.flag[env]) return;
    const Model<T>& m = models[b.obj_id[env]];
    Scratch<T, KS_LDS T*> scr{(KS_LDS T*)lds + threadIdx.x, WAVE};
    LaneState<T> st;
    T hq[4], q0[NQ];
    KS_UNROLL
    for (int i = 0; i < 4; i++) hq[i] = b.hand_quat[(long)i * N + env];
    KS_UNROLL
    for (int i = 0; i < NQ; i++) q0[i] = b.qpos0[(long)i * N + env];
    ColW<T> snap{b.snap + env, N};
    lane_reset(m, st, hq, q0, scr, snap);
    store_state(b, env, N, st);
    b.step_count[env] = 0;
    b.ncon[env] = 0;
    if (clear_flag) b.flag[env] = 0;
}

// scatter caller-provided initial states into the stored per-env initial state and flag the envs
template <typename T>
__global__ void k_store_init(Buffers<T> b, const int32_t* __restrict__ env_ids, int n, const T* __restrict__ qpos0, const T* __restrict__ hq,
                             const int32_t* __restrict__ object_id, const T* __restrict__ mass_friction, int n_models, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int env = env_ids ? env_ids[i] : i;
    if (env < 0 || env >= N) return;
    for (int k = 0; k < NQ; k++) b.qpos0[(long)k * N + env] = qpos0[(long)k * n + i];
    for (int k = 0; k < 4; k++) b.hand_quat[(long)k * N + env] = hq[(long)k * n + i];
    if (object_id) {
        int o = object_id[i];
        o = o < 0 ? 0 : (o >= n_models ? n_models - 1 : o);
        b.obj_id[env] = o;
        if (!mass_friction) { b.envp[env] = b.nominal[2 * o]; b.envp[(long)N + env] = b.nominal[2 * o + 1]; }   // the new object's own mass / friction
    }
    if (mass_friction) { b.envp[env] = mass_friction[i]; b.envp[(long)N + env] = mass_friction[(long)n + i]; }
    b.flag[env] = 1;
}

// The stepping kernel's work list: envs grouped by object model (stable: ascending env id inside a group), every group
// padded to whole workgroups with -1.  One workgroup of 256 threads; thread t owns a contiguous chunk of envs.
constexpr int SLOT_THREADS = 256, MODELS_MAX = 48;      // (48: the reference's whole object table - 42 keys - fits one context)
__global__ __launch_bounds__(SLOT_THREADS) void k_slots(const int32_t* __restrict__ obj_id, int N, int n_models, int epw, int n_wg,
                                                          int32_t* __restrict__ slot_env, int32_t* __restrict__ wg_model, int32_t* __restrict__ n_used) {
    __shared__ int cnt[SLOT_THREADS][MODELS_MAX + 1];     // +1: odd stride
    __shared__ int base[MODELS_MAX + 1];
    const int t = threadIdx.x, chunk = (N + SLOT_THREADS - 1) / SLOT_THREADS, e0 = t * chunk, e1 = e0 + chunk < N ? e0 + chunk : N;
    for (int m = 0; m < n_models; m++) cnt[t][m] = 0;
    for (int e = e0; e < e1; e++) cnt[t][obj_id[e]]++;
    for (int s = t; s < n_wg * epw; s += SLOT_THREADS) slot_env[s] = -1;
    __syncthreads();
    if (t < n_models) {                                   // exclusive scan over the chunks, per model
        int run = 0;
        for (int k = 0; k < SLOT_THREADS; k++) { const int c = cnt[k][t]; cnt[k][t] = run; run += c; }
        base[t + 1] = (run + epw - 1) / epw * epw;        // group size, whole workgroups
    }
    __syncthreads();
    if (t == 0) {
        base[0] = 0;
        for (int m = 0; m < n_models; m++) base[m + 1] += base[m];
        for (int m = 0; m < n_models; m++)
            for (int w = base[m] / epw; w < base[m + 1] / epw && w < n_wg; w++) wg_model[w] = m;
        for (int w = base[n_models] / epw; w < n_wg; w++) wg_model[w] = 0;        // idle workgroups
        *n_used = base[n_models] / epw < n_wg ? base[n_models] / epw : n_wg;      // groups that hold envs (the list is sized for one partly filled group per object)
    }
    __syncthreads();
    for (int e = e0; e < e1; e++) {
        const int m = obj_id[e], s = base[m] + cnt[t][m]++;
        if (s < n_wg * epw) slot_env[s] = e;
    }
}

// Round 6, free-running waves of a ONE-object context: the slot list re-sorted by episode step before every ks_rollout launch, so that the four envs
// of a wave are in the same phase of their episodes.  A wave's lanes run in lock step, so an env-step costs it its most expensive env - and what an
// env-step costs is mostly decided by where the env is in its episode (approach: no contact; grasp and lift: penetration queries in every substep).
// Dealt in env order a wave nearly always holds an env of the expensive phases; sorted, it pays for them only while its own envs are there
// (profiles/r06_wave_chains.txt, the model on measured counters: mean wave -7 % sorted over the whole context, -5 % sorted within the workgroups).
// Per env nothing changes: state, pair memory, noise stream and replay rows are all indexed by env id, a slot only says which lanes step the env
// (test_every_scheduling_form_of_the_rollout_kernel_equals_lock_step).
// DEFAULT (KS_ROLLOUT_PHASE_DEAL=1): every workgroup's OWN sixteen slots sorted - its envs stay the sixteen neighbours in memory they were, whose
// [field][env] columns share one 64-byte line per field.  KS_ROLLOUT_PHASE_DEAL=2 sorts the whole list: measured +2.8 % (default form) against
// env order, but a workgroup's envs are then scattered over the columns and k_rollout's FETCH_SIZE rises from 10.9 to 39.8 MB per env-step.
constexpr int PHASE_THREADS = 1024, PHASE_ENVS_MAX = 4096;
__global__ __launch_bounds__(PHASE_THREADS) void k_slots_by_phase(const int64_t* __restrict__ t, int N, int n_slots, int32_t* __restrict__ slot_env) {
    // the whole list: rank of (t, env) by counting, envs first, padding (-1) last.  One workgroup.
    __shared__ unsigned key[PHASE_ENVS_MAX];
    for (int e = threadIdx.x; e < N; e += PHASE_THREADS) {
        long long te = t[e];
        te = te < 0 ? 0 : (te > 0x7fff ? 0x7fff : te);
        key[e] = ((unsigned)te << 16) | (unsigned)e;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < N; e += PHASE_THREADS) {
        const unsigned k = key[e];
        int rank = 0;
        for (int j = 0; j < N; j++) rank += key[j] < k ? 1 : 0;
        slot_env[rank] = e;
    }
    for (int s = N + threadIdx.x; s < n_slots; s += PHASE_THREADS) slot_env[s] = -1;
}
__global__ void k_slots_by_phase_in_groups(const int64_t* __restrict__ t, int n_groups, int epw, int32_t* __restrict__ slot_env) {
    // every group of epw (<= 16) slots on its own: one thread per group, insertion sort by (t, env), padding (-1) last
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    int32_t* sl = slot_env + (long)g * epw;
    unsigned long long key[16];
    for (int i = 0; i < epw; i++) {
        const int e = sl[i];
        long long te = e >= 0 ? t[e] : 0x7fff;
        te = te < 0 ? 0 : (te > 0x7fff ? 0x7fff : te);
        key[i] = e >= 0 ? (((unsigned long long)te << 32) | (unsigned)e) : ~0ull;
    }
    for (int i = 1; i < epw; i++) {
        const unsigned long long k = key[i];
        int j = i - 1;
        while (j >= 0 && key[j] > k) { key[j + 1] = key[j]; j--; }
        key[j + 1] = k;
    }
    for (int i = 0; i < epw; i++) sl[i] = key[i] == ~0ull ? -1 : (int)(unsigned)(key[i] & 0xffffffffull);
}

// one (env, ray, geom) per lane: eight envs per wave, the eight mesh geoms of an env in adjacent lanes (lane 0 of the
// group also takes the ground plane), nearest hit by a 3-step butterfly.  The ray index is uniform per workgroup.
// pruning distance shared by the eight lanes of one (env, ray): the minimum of their nearest hits so far, published in
// LDS (lanes that have left the traversal keep their final value there; a stale read only prunes less).
template <typename T> struct GroupBound {
    KS_LDS T* group;   // the RG published distances of this (env, ray)
    int me;
    __device__ T operator()(T best) const {
        if (best >= 0 && best < group[me]) group[me] = best;
        T b = group[0];
        KS_UNROLL
        for (int k = 1; k < RG; k++) { const T o = group[k]; b = o < b ? o : b; }
        return b;
    }
};

constexpr int RAY_ENVS = WAVE / RG;
template <typename T> __global__ __launch_bounds__(WAVE) void k_rays(const Model<T>* __restrict__ models, Buffers<T> b, int N, int masked) {
    const int g = 1 + (threadIdx.x & (RG - 1));
    const int env = blockIdx.x * RAY_ENVS + (threadIdx.x >> RG_BITS), ray = blockIdx.y;
    const bool live = env < N && !(masked && !b.flag[env]);
    T best = T(-1);
    __shared__ T pub[WAVE];
    __shared__ unsigned stk[RAY_STACK * WAVE];
    pub[threadIdx.x] = Lim<T>::big;
    if (live) {
        const Model<T>& m = models[b.obj_id[env]];
        Col<T> snap{b.snap + env, N};
        T pnt[3], vec[3];
        const int sb = ray_origin(m, snap, ray, pnt, vec);
        if (g == 1) {
            best = ray_ground(m, pnt, vec);
            if (best >= 0) pub[threadIdx.x] = best;
        }
#ifdef KS_RAY_COUNT
        // diagnostic build: ray_mesh returns its node visits; they go to the contact tap buffer, rows ray * 8 + (g - 1)
        T cnt = T(0);
        if (g < m.ngeom && m.geom_body[g] != sb) {
            cnt = ray_geom(m, snap, g, pnt, vec, GroupBound<T>{(KS_LDS T*)pub + (threadIdx.x & ~(RG - 1)), (int)(threadIdx.x & (RG - 1))},
                           LdsStack<T>{(KS_LDS unsigned*)stk + threadIdx.x, WAVE});
            if (cnt < 0) cnt = T(0);
        }
        b.contact[((long)ray * RG + (g - 1)) * N + env] = cnt;
#else
        if (g < m.ngeom && m.geom_body[g] != sb)
            best = ray_nearer(best, ray_geom(m, snap, g, pnt, vec, GroupBound<T>{(KS_LDS T*)pub + (threadIdx.x & ~(RG - 1)), (int)(threadIdx.x & (RG - 1))},
                                              LdsStack<T>{(KS_LDS unsigned*)stk + threadIdx.x, WAVE}));
#endif
    }
    KS_UNROLL
    for (int mask = 1; mask < RG; mask <<= 1) best = ray_nearer(best, (T)__shfl_xor(best, mask));
    if (live && g == 1) b.rays[(long)ray * N + env] = best;
}

// phase 1 (part = -1: all of it; 0..3: that quarter of the slots, see build_obs): the observation, straight to its destination
template <typename T, typename SnapT>
__device__ __forceinline__ void obs_write(const Model<T>& m, const Buffers<T>& b, int env, int N, int mode, const ObsOut<T>& o, SnapT snap,
                                          const T* rays, int part) {
    const bool lifted0 = object_lifted(m, snap);
    uint8_t d = 0;
    const int sc = mode == 0 ? b.step_count[env] + 1 : 0;            // (the counter itself is advanced in obs_finish)
    if (mode == 0 || mode == 2) d = (lifted0 ? 1 : 0) | ((mode == 0 && o.horizon > 0 && sc >= o.horizon) ? 2 : 0);
    const bool restart = mode == 0 && d && o.auto_reset;
    // the episode is over: this observation is the terminal one (final_obs); the env restarts from its stored initial state,
    // whose observation was computed when that state was set
    T* dst = restart ? o.final_obs : o.obs;
    T* keep = mode == 1 ? b.obs0 + (long)env * NOBS : nullptr;
    const long base = o.env_major ? (long)env * NOBS : (long)env, stride = o.env_major ? 1 : (long)N;
    T rew, inf[3];
    bool lifted;
    build_obs(m, snap, rays, [&](int j, T v) {
        if (dst) dst[base + j * stride] = v;
        if (keep) keep[j] = v;
    }, rew, lifted, inf, part);
}

// phase 2 (after every part of phase 1 has read the snapshot): counters, reward, flags, and the restart of a finished episode
template <typename T, typename SnapT, typename ScrT>
__device__ __forceinline__ void obs_finish(const Model<T>& m, const Buffers<T>& b, int env, int N, int mode, const ObsOut<T>& o, SnapT snap, ScrT rscr) {
    // one decision for done, reward and the destination: object_lifted's (the same arithmetic as build_obs' own test; this only
    // rules out two inlined copies ever being contracted differently by the compiler)
    const bool lifted0 = object_lifted(m, snap);
    uint8_t d = 0;
    int sc = 0;
    if (mode == 0) { sc = b.step_count[env] + 1; b.step_count[env] = sc; }
    if (mode == 0 || mode == 2) d = (lifted0 ? 1 : 0) | ((mode == 0 && o.horizon > 0 && sc >= o.horizon) ? 2 : 0);
    const bool restart = mode == 0 && d && o.auto_reset;
    const long base = o.env_major ? (long)env * NOBS : (long)env, stride = o.env_major ? 1 : (long)N;
    const T rew = lifted0 ? T(50) : T(0);
    if (mode == 0 || mode == 2) {
        if (o.reward) o.reward[env] = rew;
        if (o.done) o.done[env] = d;
        if (o.info) { o.info[env] = T(0); o.info[(long)N + env] = T(0); o.info[2L * N + env] = rew; }
    }
    if (restart) {
        if (o.obs) {
            for (int j = 0; j < NOBS; j++) o.obs[base + j * stride] = b.obs0[(long)env * NOBS + j];
        }
        // ... and the restart itself (what k_reset does for a caller's ks_reset): state, snapshot, counters
        LaneState<T> st;
        T hq[4], q0[NQ];
        KS_UNROLL
        for (int i = 0; i < 4; i++) hq[i] = b.hand_quat[(long)i * N + env];
        KS_UNROLL
        for (int i = 0; i < NQ; i++) q0[i] = b.qpos0[(long)i * N + env];
        ColW<T> rsnap{b.snap + env, N};
        lane_reset(m, st, hq, q0, rscr, rsnap);
        store_state(b, env, N, st);
        b.step_count[env] = 0;
        b.ncon[env] = 0;
    }
}

// One env's episode bookkeeping + observation.
// mode 0: after a step (reward / done / time limit / auto-reset flagging); mode 1: after a reset
// (observation of flagged envs only, flag cleared); mode 2: observation / reward / lifted flag of whatever snapshot and
// rays the buffers hold, no episode bookkeeping (ks_obs_from_snapshot, the parity hook for the env-layer golden vectors).
// `snap` reads the env's snapshot, `rays` its 17 distances, `rscr` is scratch for an auto-reset's forward kinematics (the
// body-pose part of a stepping block, SCR_CON words).  The observation is written straight to where it belongs (the
// termination test runs first), nothing of it is kept in registers.
template <typename T, typename SnapT, typename ScrT>
__device__ __forceinline__ void obs_epilogue(const Model<T>& m, const Buffers<T>& b, int env, int N, int mode, const ObsOut<T>& o, SnapT snap,
                                             const T* rays, ScrT rscr) {
    obs_write(m, b, env, N, mode, o, snap, rays, -1);
    obs_finish(m, b, env, N, mode, o, snap, rscr);
}

template <typename T>
__global__ __launch_bounds__(WAVE) void k_obs(const Model<T>* __restrict__ models, Buffers<T> b, int N, int mode, ObsOut<T> o) {
    __shared__ T rlds[SCR_CON * WAVE];      // scratch of an auto-reset's forward kinematics (body-pose part only, as in k_reset)
    const int env = blockIdx.x * WAVE + threadIdx.x;
    if (env >= N) return;
    if (mode == 1) {
        if (!b.flag[env]) return;
        b.flag[env] = 0;
    }
    Col<T> snap{b.snap + env, N};
    T rays[NRAY];
    KS_UNROLL
    for (int i = 0; i < NRAY; i++) rays[i] = b.rays[(long)i * N + env];
    obs_epilogue(models[b.obj_id[env]], b, env, N, mode, o, snap, rays, Scratch<T, KS_LDS T*>{(KS_LDS T*)rlds + threadIdx.x, WAVE});
}

// The tail of the stepping kernel (fp32 / LDS variant, after the rays): the workgroup finishes its envs' step - the 82-d
// observation (four waves, a quarter of the slots each), then per env the time limit, termination, reward and the restart of
// a finished episode - from the env's snapshot and ray distances in global memory (whoever cast the rays).  What leaves the
// critical path against a k_obs launch is the launch, its gap and the wait for the slowest workgroup before ANY env's
// observation could start.
template <bool WAVE_SCOPE>
__device__ __noinline__ void wg_obs(const Model<float>& m, const Buffers<float>& b, int N, int slot0, int epw, KS_LDS unsigned* w, const ObsOut<float>& o) {
    using C = Crew<WAVE_SCOPE>;
    KS_LDS float* rscr = (KS_LDS float*)(w + wg_rays_words(epw, C::NTH) + epw * (NRAY + 1));
    // team-parallel: wave p of the workgroup writes part p of the 82 slots (build_obs) of env e = lane, for the 16 envs at once -
    // four instruction streams of a quarter of the length on the four SIMDs; then one thread per env finishes the step.
    // (One wave on its own: lane p of env e's DPP row takes part p - the four parts run one after the other, the wave's envs side by side.)
    const int e = WAVE_SCOPE ? (int)(threadIdx.x & 63) >> 4 : (int)(threadIdx.x & 63), part = WAVE_SCOPE ? (int)(threadIdx.x & 15) : (int)(threadIdx.x >> 6);
    const int env = e < epw ? b.slot_env[slot0 + e] : -1;
    if (env >= 0 && part < 4) {
        float rays[NRAY];
        KS_UNROLL
        for (int r = 0; r < NRAY; r++) rays[r] = b.rays[(long)r * N + env];
        obs_write(m, b, env, N, 0, o, Col<float>{b.snap + env, N}, rays, part);
    }
    C::sync();                                          // every part has read the snapshot (a restart overwrites it)
    if (env >= 0 && part == 0)
        obs_finish(m, b, env, N, 0, o, Col<float>{b.snap + env, N}, ScratchC<float, KS_LDS float*>{rscr + e * (SCR_CON + 1)});
}

__device__ __forceinline__ int pool_wait(const int32_t* p, bool& ok) {
    const long long t0 = wall_clock64();
    int v;
    while ((v = __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
        if (wall_clock64() - t0 > 50000000ll) { ok = false; break; }       // 0.5 s of the 100 MHz clock (a real wait is < 1 ms): report, do not hang
        __builtin_amdgcn_s_sleep(8);
    }
    return v;
}

// The ray pool of a stepping launch (see k_env_step).  A workgroup's ticket (its 16 envs' rays: wg_rays with all 256 threads) is
// published one substep before the workgroup is done and has a state word: 1 published, 2 claimed, 3 done.
//   * A workgroup that finishes stepping first claims its OWN ticket (a compare-and-swap on its state word) and casts its own
//     rays, as it always did; early finishers then leave, and their CUs' registers go to the learner's waves.
//   * A workgroup that finishes when only the last POOL_LINGER_LAST tickets of the launch are still unpublished stays: it takes
//     positions of that last stretch of the queue in order (one fetch-add each - no contended compare-and-swap loops: 256
//     workgroups hammering one word cost 0.15 ms), waits for the ticket at its position to be published, claims it unless its
//     owner was quicker, and casts its rays.  So when the slowest workgroups publish, hands are free at that moment and their
//     rays are cast while they compute their last substep (~60 us) instead of behind it.
//   * Everybody finally waits for its own ticket to be done (by itself or by a helper).
// A workgroup only stays when every workgroup of the launch has STARTED (a counter at kernel entry): with more workgroups than
// CUs (8192 envs: two rounds) a helper waiting for a ticket of a workgroup that cannot start until the helper leaves would
// never see it.
__device__ __noinline__ void wg_ray_pool(const Model<float>* models, const Model<float>& mine, const Buffers<float>& b, int N, int epw, KS_LDS unsigned* w,
                                         int n_wg, int linger) {
    int32_t* q = b.rayq;
    int32_t* state = q + 4 + n_wg;
    KS_LDS int* bc = (KS_LDS int*)(w + wg_rays_words(EPW_MAX) + wg_obs_words(EPW_MAX));      // thread 0's decision, for everybody
    const int own = blockIdx.x, first_late = n_wg > POOL_LINGER_LAST ? n_wg - POOL_LINGER_LAST : 0;
    bool ok = true, late = false, tried_own = false;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = -1;
            if (!tried_own) {
                tried_own = true;
                // (every workgroup of the launch has started: nobody we might wait for is still queued behind the CUs we hold)
                late = linger && __hip_atomic_load(&q[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= first_late &&
                       __hip_atomic_load(&q[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_wg;
                if (atomicCAS(&state[own], 1, 2) == 1) t = own;
            }
            while (t < 0 && late) {
                const int c = first_late + atomicAdd(&q[1], 1);
                if (c >= n_wg) break;
                const int cand = pool_wait(&q[4 + c], ok) - 1;
                if (cand < 0) break;
                if (atomicCAS(&state[cand], 1, 2) == 1) t = cand;
            }
            *bc = t;
        }
        __syncthreads();
        const int t = *bc;
        if (t < 0) break;
        // (the same object as mine: my LDS copy of the model - a flat load that resolves to LDS costs a fraction of one that goes to L2)
        const int tm = b.wg_model[t];
        wg_rays<false>(tm == b.wg_model[own] ? mine : models[tm], b, N, t * epw, epw, w KS_RAY_PROF_ARG(nullptr));
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&state[t], 3, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    // my own envs' rays: cast by me or by a helper (the acquire also drops this CU's stale lines)
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(&state[own], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 3) {
            if (wall_clock64() - t0 > 50000000ll) { ok = false; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        *bc = ok ? 1 : 0;
    }
    __syncthreads();
    if (*bc == 0 && (int)threadIdx.x < epw) {
        const int env = b.slot_env[own * epw + threadIdx.x];
        if (env >= 0) atomicOr(&b.status[env], ST_RAY_POOL_TIMEOUT);
    }
    __syncthreads();
}

struct CtxBase {
    ks_config cfg;
    int device;
    std::string error;
    bool model_loaded = false;
    virtual ~CtxBase() {}
    virtual int load_models(int n_models, const void* const* blobs, const size_t* sizes) = 0;
    virtual int reset(const int32_t* ids, int n, const void* q0, const void* hq, const int32_t* object_id, const void* mass_friction, void* obs,
                      hipStream_t s) = 0;
    virtual int step(const void* action, void* obs, void* reward, uint8_t* done, void* info, void* final_obs, hipStream_t s) = 0;
    virtual int get_state(void* qpos, void* qvel, void* warm, void* contact, int32_t* ncon, int32_t* status, hipStream_t s) = 0;
    virtual int set_state(const void* qpos, const void* qvel, const void* warm, hipStream_t s) = 0;
    virtual int set_env_params(const void* mass, const void* mu, hipStream_t s) = 0;
    virtual int substep(const void* ctrl, hipStream_t s) = 0;
    virtual int obs_from_snapshot(const void* snap, const void* rays, void* obs, void* reward, uint8_t* done, void* info, hipStream_t s) = 0;
    virtual int kernel_time(int reset, double* avg_ms, int64_t* launches) = 0;
    virtual int rollout(int n_iter, const ks_rollout_args* args, hipStream_t s) = 0;
    virtual int rollout_plan(int32_t* mode, int32_t* groups, int32_t* workgroups) = 0;
};

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            error = std::string(#expr) + ": " + hipGetErrorString(e_);                       \
            return KS_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

template <typename T> struct Ctx : CtxBase {
    static constexpr bool USE_LDS = sizeof(T) == 4;
    Buffers<T> b{};
    Buffers<T>* d_b = nullptr;            // the same pointer table in device memory (what the stepping kernels' out-of-line tails read)
    ks_rollout_args* d_ra = nullptr;      // ks_rollout's argument record in device memory (+ a pinned host ring to copy it from)
    ks_rollout_args* h_ra = nullptr;
    unsigned h_ra_next = 0;
    ObsOut<T>* d_out = nullptr;           // ... and where ks_step's results go (re-sent only when a caller changes its buffers)
    ObsOut<T> out_sent{};
    ObsOut<T>* h_out = nullptr;           // pinned staging of that record (an async copy from pinned memory may be captured in a
    int h_out_next = 0;                   // graph): a ring, so that a change does not overwrite a copy that is still queued
    static constexpr int H_OUT_RING = 16;
    bool out_valid = false;
    // A CAPTURED copy reads its pinned source at every replay of the graph, long after this call: captured calls therefore get a
    // record of their own that is never recycled (freed with the context), not a slot of the rings above.
    // (allocated with the rings: no allocation is allowed while a stream captures; CAPTURE_RECORDS captured calls per context)
    static constexpr int CAPTURE_RECORDS = 128;
    int captured_out = 0, captured_ra = 0;
    template <typename R> R* pinned_record(bool capturing, R* ring, unsigned slot_index, int& used) {
        if (!capturing) return ring + (slot_index % H_OUT_RING);
        if (used >= CAPTURE_RECORDS) return nullptr;
        return ring + H_OUT_RING + used++;
    }
    static bool stream_is_capturing(hipStream_t s) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        return cap != hipStreamCaptureStatusNone;
    }
    Model<T>* d_model = nullptr;          // [n_models] model table
    int n_models = 0, n_wg = 0;
    int32_t n_groups = 0;                 // groups of the slot list that hold envs (<= n_wg): what ks_rollout deals
    std::map<std::pair<size_t, uint64_t>, void*> shared;      // uploaded arrays by (bytes, content hash): the hand's meshes are
                                                              // the same in every object's blob and are kept once
    std::vector<void*> allocs;
    // HIP-event timing of k_env_step
    static constexpr int NEV = 512;
    int ev_stride = 4;                    // KS_EVENT_STRIDE
    long step_calls = 0;
    std::vector<hipEvent_t> ev0, ev1;
    int ev_used = 0;
    double ev_ms = 0;
    int64_t ev_launches = 0;

    template <typename U> int alloc(U** p, size_t count) {
        HIPCHK(hipMalloc((void**)p, count * sizeof(U)));
        HIPCHK(hipMemset(*p, 0, count * sizeof(U)));
        allocs.push_back(*p);
        return KS_OK;
    }
    int init() {
        const size_t N = cfg.n_envs;
        int r;
        if (getenv("KS_PAIR_MEMORY")) cfg.pair_memory = getenv("KS_PAIR_MEMORY")[0] != '0';      // experiment switch
        if (getenv("KS_EVENT_STRIDE")) ev_stride = atoi(getenv("KS_EVENT_STRIDE")) > 0 ? atoi(getenv("KS_EVENT_STRIDE")) : 1;
        if ((r = alloc(&b.qpos, NQ * N))) return r;
        if ((r = alloc(&b.qvel, NV * N))) return r;
        if ((r = alloc(&b.warm, NV * N))) return r;
        if ((r = alloc(&b.hand_quat, 4 * N))) return r;
        if ((r = alloc(&b.qpos0, NQ * N))) return r;
        if ((r = alloc(&b.snap, SNAP_TOTAL * N))) return r;
        if ((r = alloc(&b.rays, NRAY * N))) return r;
        if ((r = alloc(&b.obs0, (size_t)NOBS * N))) return r;
        if ((r = alloc(&b.contact, (size_t)NCON_MAX * CON_STRIDE * N))) return r;
        if (!USE_LDS && (r = alloc(&b.gscratch, (size_t)SCR_TOTAL * N))) return r;
        if ((r = alloc(&b.envp, (size_t)2 * N))) return r;
        if (USE_LDS && (r = alloc(&b.pairmem, (size_t)SUBS * ((NPAIR_MAX + SUBS - 1) / SUBS) * WARM_WORDS * N))) return r;
        if ((r = alloc(&b.ncon, N))) return r;
        if ((r = alloc(&b.status, N))) return r;
        if ((r = alloc(&b.step_count, N))) return r;
        if ((r = alloc(&b.flag, N))) return r;
        if ((r = alloc(&b.obj_id, N))) return r;
        ev0.resize(NEV); ev1.resize(NEV);
        for (int i = 0; i < NEV; i++) { HIPCHK(hipEventCreate(&ev0[i])); HIPCHK(hipEventCreate(&ev1[i])); }
        return KS_OK;
    }
    ~Ctx() override {
        for (void* p : allocs) (void)hipFree(p);
        if (h_out) (void)hipHostFree(h_out);
        if (h_ra) (void)hipHostFree(h_ra);
        if (d_ra) (void)hipFree(d_ra);
        if (d_queue) (void)hipFree(d_queue);
        for (auto& e : ev0) (void)hipEventDestroy(e);
        for (auto& e : ev1) (void)hipEventDestroy(e);
    }
    // device copy of a host array; identical content (the hand's hull / ray meshes, shared by all object blobs) is uploaded once
    template <typename U> int upload(const std::vector<U>& v, const U** out) {
        const size_t bytes = v.size() * sizeof(U);
        uint64_t h = 1469598103934665603ull;
        const unsigned char* p = (const unsigned char*)v.data();
        for (size_t i = 0; i < bytes; i++) { h ^= p[i]; h *= 1099511628211ull; }
        auto it = shared.find({bytes, h});
        if (it != shared.end()) { *out = (const U*)it->second; return KS_OK; }
        U* d = nullptr;
        int r = alloc(&d, v.size() ? v.size() : 1);
        if (r) return r;
        HIPCHK(hipMemcpy(d, v.data(), bytes, hipMemcpyHostToDevice));
        shared[{bytes, h}] = d;
        *out = d;
        return KS_OK;
    }
    int load_models(int nm, const void* const* blobs, const size_t* sizes) override {
        if (model_loaded) { error = "ks_load_model: a context loads its model(s) once"; return KS_ERR_STATE; }
        if (nm <= 0 || nm > MODELS_MAX) { error = "ks_load_models: between 1 and 48 object models"; return KS_ERR_INVALID; }
        std::vector<Model<T>> table(nm);
        std::vector<T> nominal(2 * (size_t)nm);
        hull_words = 0;
        for (int k = 0; k < nm; k++) {
            HostModel<T> hm;
            if (!parse_model<T>(blobs[k], sizes[k], hm)) { error = "ks_load_model: " + hm.error; return KS_ERR_MODEL; }
            // the stepping kernel gives every lane of an env's team at most HPL hull pairs (ks_core.h, collision)
            if (hull_pair_count(hm.m) > HPL * SUBS || hm.m.npair - hull_pair_count(hm.m) > 32) { error = "ks_load_model: too many contact pairs for this build"; return KS_ERR_MODEL; }
            int words = 0, adj_ints = 0, r;
            for (int s = 0; s < hm.m.nmesh; s++) {
                if ((r = upload(hm.vert[s], &hm.m.mesh_vert[s]))) return r;
                if ((r = upload(hm.tri[s], &hm.m.mesh_tri[s]))) return r;
                if ((r = upload(hm.bvh_box[s], &hm.m.mesh_bvh_box[s]))) return r;
                if ((r = upload(hm.bvh_lr[s], &hm.m.mesh_bvh_lr[s]))) return r;
                if ((r = upload(hm.adj_off[s], &hm.m.mesh_adj_off[s]))) return r;
                if ((r = upload(hm.adj[s], &hm.m.mesh_adj[s]))) return r;
                words += hm.m.mesh_nvert_pad[s] * 4;
                adj_ints += ((hm.m.mesh_nvert[s] + 1 + 3) & ~3) + hm.m.mesh_nchunk[s] * 4;
            }
            if ((r = upload(hm.dirtab, &hm.m.mesh_dirtab))) return r;
            if (!MULTI_GEOM) {
                // the LDS image of the hull tables (stage_hulls' layout) as one block
                std::vector<unsigned char> pack;
                auto put = [&](const void* p, size_t bytes) { pack.insert(pack.end(), (const unsigned char*)p, (const unsigned char*)p + bytes); };
                for (int s = 0; s < hm.m.nmesh; s++) put(hm.vert[s].data(), hm.vert[s].size() * sizeof(T));
                for (int s = 0; s < hm.m.nmesh; s++) {
                    put(hm.adj_off[s].data(), hm.adj_off[s].size() * sizeof(unsigned short));
                    put(hm.adj[s].data(), hm.adj[s].size() * sizeof(unsigned short));
                }
                pack.resize((pack.size() + 15) / 16 * 16, 0);
                const unsigned char* d = nullptr;
                if ((r = upload(pack, &d))) return r;
                hm.m.hull_pack = d;
                hm.m.hull_pack_bytes = (int)pack.size();
            } else {
                words = 0; adj_ints = 0;          // the tables stay in global memory: only the pair records take LDS
            }
            const int iwords = (adj_ints * (int)sizeof(unsigned short) + (int)sizeof(T) - 1) / (int)sizeof(T);
            words += (iwords + 3) & ~3;
            words += NPAIR_MAX * pair_rec_bytes<T>() / (int)sizeof(T);
            hull_words = words > hull_words ? words : hull_words;          // LDS is sized for the largest object
            table[k] = hm.m;
            nominal_env_params(hm.m, nominal[2 * k], nominal[2 * k + 1]);
        }
        n_models = nm;
        int r;
        if ((r = alloc(&d_model, (size_t)nm))) return r;
        HIPCHK(hipMemcpy(d_model, table.data(), sizeof(Model<T>) * nm, hipMemcpyHostToDevice));
        if ((r = alloc(&b.nominal, nominal.size()))) return r;
        HIPCHK(hipMemcpy(b.nominal, nominal.data(), nominal.size() * sizeof(T), hipMemcpyHostToDevice));
        {
            // every env starts with object 0 and its nominal parameters until ks_reset_objects / ks_set_env_params say otherwise
            std::vector<T> ep(2 * (size_t)cfg.n_envs, nominal[1]);
            std::fill(ep.begin(), ep.begin() + cfg.n_envs, nominal[0]);
            HIPCHK(hipMemcpy(b.envp, ep.data(), ep.size() * sizeof(T), hipMemcpyHostToDevice));
        }
        if ((r = plan_launch()) != KS_OK) return r;
        // the stepping kernel's work list: at most one partly filled workgroup per object
        n_wg = (cfg.n_envs + lpw - 1) / lpw + (nm - 1);
        if ((r = alloc(&b.slot_env, (size_t)n_wg * lpw))) return r;
        if ((r = alloc(&b.wg_model, (size_t)n_wg))) return r;
        if ((r = alloc(&b.n_used, (size_t)1))) return r;
        if ((r = alloc(&b.rayq, (size_t)4 + 2 * (size_t)n_wg))) return r;
        // the rays pooled over the launch (wg_ray_pool); KS_RAY_POOL=0 makes every workgroup cast its own envs' rays
        if (obs_in_step && cfg.frame_skip >= 1 && !(getenv("KS_RAY_POOL") && getenv("KS_RAY_POOL")[0] == '0')) {
            ray_pool = 2;
        }
        hipLaunchKernelGGL(k_slots, dim3(1), dim3(SLOT_THREADS), 0, 0, b.obj_id, cfg.n_envs, n_models, lpw, n_wg, b.slot_env, b.wg_model, b.n_used);
        HIPCHK(hipGetLastError());
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(&n_groups, b.n_used, sizeof(int32_t), hipMemcpyDeviceToHost));
        if ((r = alloc(&d_b, (size_t)1))) return r;
        if ((r = alloc(&d_out, (size_t)1))) return r;
        if (!h_out) HIPCHK(hipHostMalloc((void**)&h_out, (H_OUT_RING + CAPTURE_RECORDS) * sizeof(ObsOut<T>), hipHostMallocDefault));
        if (!d_ra) HIPCHK(hipMalloc((void**)&d_ra, sizeof(ks_rollout_args)));
        if (!d_queue) HIPCHK(hipMalloc((void**)&d_queue, ROLLOUT_Q_WORDS * sizeof(int)));        // (not at launch time: ks_rollout may be captured)
        if (!h_ra) HIPCHK(hipHostMalloc((void**)&h_ra, sizeof(ks_rollout_args) * (H_OUT_RING + CAPTURE_RECORDS), hipHostMallocDefault));
        HIPCHK(hipMemcpy(d_b, &b, sizeof b, hipMemcpyHostToDevice));
        model_loaded = true;
        return KS_OK;
    }
    int blocks() const { return (cfg.n_envs + WAVE - 1) / WAVE; }
    // envs per wave and dynamic LDS bytes of the stepping kernels
    int resident_wgs = 256;
    int* d_queue = nullptr;               // k_rollout's ready queue (ROLLOUT_Q_WORDS ints), used when the context has more groups than resident workgroups
    bool rollout_queue = !MULTI_GEOM;     // KS_ROLLOUT_DEAL=static / rr: the fixed deals of rounds 3-4 instead.  Off in the multi-geom build: its hull tables
                                          // live in global memory, L2-resident, and the queue's per-task agent-scope acquire (buffer_inv sc1) throws them out of
                                          // the XCD's L2 every time ANY of its 32 workgroups takes a group: the 14-key stage context 0.87 -> 0.44 M env-steps/s
                                          // (measured, round 5).  That context is bound by its slowest groups' SEQUENTIAL env-steps anyway (every env does the
                                          // same number of env-steps per launch): no dealing helps it.  KS_ROLLOUT_DEAL=queue forces the queue.
    bool rollout_waves = !(getenv("KS_ROLLOUT_WAVES") && getenv("KS_ROLLOUT_WAVES")[0] == '0');      // free-running waves (one group per workgroup)
    int rollout_phase_deal = getenv("KS_ROLLOUT_PHASE_DEAL") ? atoi(getenv("KS_ROLLOUT_PHASE_DEAL")) : 1;      // free waves, one object: slots sorted by episode step before every launch (1: within every workgroup's sixteen, 2: the whole list, 0: env order)
    bool rollout_round_robin = false;     // how k_rollout deals the env groups to its persistent workgroups: contiguous runs (default) or round-robin (KS_ROLLOUT_DEAL=rr)
    int lpw = WAVE;
    bool rays_in_step = false, obs_in_step = false;
    int ray_pool = 0;                     // 0 every workgroup casts its own envs' rays, 1 pooled, 2 pooled + early finishers linger
    size_t step_lds = 0;
    int hull_words = 0;
    int plan_launch() {
        {
            // workgroups of the persistent rollout kernel that are resident at once: one per compute unit (each takes all of a CU's LDS)
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
            if (const char* e = getenv("KS_ROLLOUT_WGS")) { const int v = atoi(e); if (v > 0) cus = v; }
            resident_wgs = cus;
            // mixed-object contexts: round-robin (cheap and expensive objects average out over a workgroup's groups: BASELINE config 5 with the
            // trained policy 2.89 M env-steps/s against 2.02 M with contiguous runs, round 4); one object: contiguous (nothing to restage)
            rollout_round_robin = n_models > 1;
            if (const char* e = getenv("KS_ROLLOUT_DEAL")) { rollout_round_robin = strcmp(e, "rr") == 0; rollout_queue = strcmp(e, "queue") == 0; }
        }
        const size_t lds_max = 160 * 1024;
        const size_t hull_bytes = (size_t)hull_words * sizeof(T) + (USE_LDS ? (size_t)model_words<T>() * sizeof(T) : 0);
        const size_t per_env = USE_LDS ? (size_t)SCR_TOTAL * sizeof(T) : 0;
        int cap = USE_LDS ? (int)((lds_max - hull_bytes - 64) / per_env) : EPW_MAX;
        if (cap > EPW_MAX) cap = EPW_MAX;
        if (cap < 1) { error = "hull tables do not fit in LDS"; return KS_ERR_MODEL; }
        int want = cfg.envs_per_wave > 0 ? cfg.envs_per_wave : EPW_MAX;
        lpw = want > cap ? cap : want;
        step_lds = hull_bytes + per_env * lpw;
        // fp32: the workgroups cast their own envs' rays at the end of the stepping kernel (wg_rays) when the dead env blocks
        // are large enough for its list and stacks; KS_RAYS_IN_STEP=0 keeps the separate k_rays launch
        rays_in_step = USE_LDS && sizeof(T) == 4 && (size_t)wg_rays_words(lpw) <= (size_t)SCR_TOTAL * lpw &&
                       !(getenv("KS_RAYS_IN_STEP") && getenv("KS_RAYS_IN_STEP")[0] == '0');
        // ... and then finish the step there as well (wg_obs: observation, reward, done, auto-reset); KS_OBS_IN_STEP=0 keeps k_obs
        obs_in_step = rays_in_step && (size_t)(wg_rays_words(EPW_MAX) + wg_obs_words(EPW_MAX) + 4) <= (size_t)SCR_TOTAL * lpw &&
                      !(getenv("KS_OBS_IN_STEP") && getenv("KS_OBS_IN_STEP")[0] == '0');
        if (getenv("KS_DEBUG")) fprintf(stderr, "[ks] stepping kernel: %d envs per workgroup, LDS %zu B (tables %zu B, %zu B per env), limit %zu\n", lpw, step_lds, hull_bytes, per_env, lds_max);
        if constexpr (sizeof(T) == 4) HIPCHK(hipFuncSetAttribute((const void*)k_env_step_f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_lds));
        else HIPCHK(hipFuncSetAttribute((const void*)k_env_step<T, USE_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_substep<T, USE_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_lds));
        return KS_OK;
    }
    int post_reset(void* obs, hipStream_t s) {
        const int N = cfg.n_envs;
        if (b.rayq) HIPCHK(hipMemsetAsync(b.rayq, 0, (4 + 2 * (size_t)n_wg) * sizeof(int32_t), s));    // (clean even after an aborted launch)
        hipLaunchKernelGGL((k_reset<T, USE_LDS>), dim3(blocks()), dim3(WAVE), 0, s, d_model, b, N, 0);
        hipLaunchKernelGGL((k_rays<T>), dim3((N + RAY_ENVS - 1) / RAY_ENVS, NRAY), dim3(WAVE), 0, s, d_model, b, N, 1);
        hipLaunchKernelGGL((k_obs<T>), dim3(blocks()), dim3(WAVE), 0, s, d_model, b, N, 1,
                           ObsOut<T>{(T*)obs, nullptr, nullptr, nullptr, nullptr, cfg.horizon, cfg.auto_reset, cfg.obs_env_major});
        HIPCHK(hipGetLastError());
        return KS_OK;
    }
    int reset(const int32_t* ids, int n, const void* q0, const void* hq, const int32_t* object_id, const void* mass_friction, void* obs,
              hipStream_t s) override {
        if (!model_loaded) { error = "ks_reset before ks_load_model"; return KS_ERR_STATE; }
        if (n <= 0 || n > cfg.n_envs || !q0 || !hq || (!ids && n != cfg.n_envs)) { error = "ks_reset: bad arguments"; return KS_ERR_INVALID; }
        hipLaunchKernelGGL((k_store_init<T>), dim3((n + 255) / 256), dim3(256), 0, s, b, ids, n, (const T*)q0, (const T*)hq, object_id,
                           (const T*)mass_friction, n_models, cfg.n_envs);
        // objects changed: regroup the stepping kernel's work list by object
        if (object_id && n_models > 1)
        {
            hipLaunchKernelGGL(k_slots, dim3(1), dim3(SLOT_THREADS), 0, s, b.obj_id, cfg.n_envs, n_models, lpw, n_wg, b.slot_env, b.wg_model, b.n_used);
            // how many groups hold envs decides how ks_rollout schedules them (one group per workgroup: free waves): read it back - a reset that
            // changes objects is not a hot path; under stream capture the conservative count (every slot of the list) stands
            if (!stream_is_capturing(s)) {
                HIPCHK(hipMemcpyAsync(&n_groups, b.n_used, sizeof(int32_t), hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
            } else n_groups = n_wg;
        }
        return post_reset(obs, s);
    }
    int step(const void* action, void* obs, void* reward, uint8_t* done, void* info, void* final_obs, hipStream_t s) override {
        if (!model_loaded) { error = "ks_step before ks_load_model"; return KS_ERR_STATE; }
        if (!action) { error = "ks_step: action is NULL"; return KS_ERR_INVALID; }
        const int N = cfg.n_envs;
        // ks_kernel_time's samples: every ev_stride-th launch is bracketed by two events (each costs ~4 us of stream time)
        const bool timed = ev_used < NEV && (step_calls++ % ev_stride) == 0;
        const ObsOut<T> out{(T*)obs, (T*)reward, done, (T*)info, (T*)final_obs, cfg.horizon, cfg.auto_reset, cfg.obs_env_major};
        const bool same = out.obs == out_sent.obs && out.reward == out_sent.reward && out.done == out_sent.done && out.info == out_sent.info &&
                          out.final_obs == out_sent.final_obs && out.horizon == out_sent.horizon && out.auto_reset == out_sent.auto_reset &&
                          out.env_major == out_sent.env_major;
        // Under stream capture the copy below is only RECORDED (it runs at every replay of the graph, not now): the device copy
        // cannot be taken as current afterwards, so a capturing call always records the copy and never marks it as sent.
        // Once the context holds ANY captured record (captured_out > 0) a replayed graph may have overwritten d_out with its own
        // pinned record since the last eager call - replays are invisible here - so the "same pointers as last time" shortcut is
        // off for good and every eager call re-sends its record (eager + captured calls with different buffers).
        const bool capturing = stream_is_capturing(s);
        if (obs_in_step && (capturing || captured_out > 0 || !(same && out_valid))) {
            out_valid = !capturing;
            out_sent = out;
            ObsOut<T>* slot = pinned_record(capturing, h_out, (unsigned)h_out_next++, captured_out);
            if (!slot) { error = "ks_step: more captured calls than the context keeps output records for"; return KS_ERR_STATE; }
            *slot = out;
            HIPCHK(hipMemcpyAsync(d_out, slot, sizeof out, hipMemcpyHostToDevice, s));
        }
        if (timed) HIPCHK(hipEventRecord(ev0[ev_used], s));
        if constexpr (sizeof(T) == 4)
            hipLaunchKernelGGL(k_env_step_f32, dim3(n_wg), dim3(WG), step_lds, s, d_model, b, (const Buffers<T>*)d_b, (const T*)action, N,
                               cfg.frame_skip, cfg.solver_iterations, lpw, cfg.contact_tap, (int)rays_in_step, (int)(USE_LDS && cfg.pair_memory),
                               (int)obs_in_step, (const ObsOut<T>*)d_out, ray_pool, n_wg);
        else
            hipLaunchKernelGGL((k_env_step<T, USE_LDS>), dim3(n_wg), dim3(WG), step_lds, s, d_model, b, (const Buffers<T>*)d_b, (const T*)action, N,
                               cfg.frame_skip, cfg.solver_iterations, lpw, cfg.contact_tap, (int)rays_in_step, (int)(USE_LDS && cfg.pair_memory),
                               (int)obs_in_step, (const ObsOut<T>*)d_out, ray_pool, n_wg);
        if (timed) { HIPCHK(hipEventRecord(ev1[ev_used], s)); ev_used++; }
        if (!rays_in_step) hipLaunchKernelGGL((k_rays<T>), dim3((N + RAY_ENVS - 1) / RAY_ENVS, NRAY), dim3(WAVE), 0, s, d_model, b, N, 0);
        if (!obs_in_step) hipLaunchKernelGGL((k_obs<T>), dim3(blocks()), dim3(WAVE), 0, s, d_model, b, N, 0, out);
        // (auto-reset: wg_obs / k_obs restarts finished envs from their stored initial state and returns the cached observation)
        HIPCHK(hipGetLastError());
        return KS_OK;
    }
    int rollout(int n_iter, const ks_rollout_args* ra, hipStream_t s) override {
        if (!model_loaded) { error = "ks_rollout before ks_load_model"; return KS_ERR_STATE; }
        if constexpr (sizeof(T) != 4) { error = "ks_rollout: fp32 contexts only"; return KS_ERR_INVALID; }
        else {
            if (!ra || n_iter <= 0 || !ra->actor_pub || !ra->actor_ver || !ra->obs || !ra->prev_obs || !ra->has_prev || !ra->ready || !ra->lifting ||
                !ra->t || !ra->steps_total || !ra->action || !ra->action_t || !ra->reward_out || !ra->done_out || !ra->sim_obs || !ra->sim_reward ||
                !ra->sim_done || !ra->sim_info || !ra->sim_final_obs || !ra->counters) { error = "ks_rollout: NULL argument"; return KS_ERR_INVALID; }
            if (ra->with_replay && (!ra->cur_state || !ra->cur_next || !ra->cur_action || !ra->cur_reward || !ra->cur_not_done || !ra->cur_len ||
                                    !ra->cur_sel || !ra->pub_len || ra->horizon <= ra->n_steps)) { error = "ks_rollout: replay buffers"; return KS_ERR_INVALID; }
            if (!obs_in_step || !cfg.obs_env_major || !cfg.auto_reset) { error = "ks_rollout needs the in-kernel observation path, env-major observations and auto_reset"; return KS_ERR_STATE; }
            if ((ra->off_w2 | ra->off_w3 | ra->actor_stride) & 3) { error = "ks_rollout: weight offsets must be multiples of 4 floats"; return KS_ERR_INVALID; }
            if ((ra->h1 | ra->h2) & 3) { error = "ks_rollout: hidden widths must be multiples of 4"; return KS_ERR_INVALID; }
            if ((size_t)(((ra->h1 + 15) / 16 + (ra->h2 + 15) / 16) * 4 + 4) * 16 * 16 + 16 > (size_t)SCR_TOTAL * lpw * sizeof(T)) { error = "ks_rollout: no LDS for the policy"; return KS_ERR_STATE; }
            const int N = cfg.n_envs;
            if (ra->budget_ticks != 0 && (ra->budget_ticks < 0 || !(plan_waves() && (size_t)(((ra->h1 + 15) / 16 + (ra->h2 + 15) / 16) * 4) * 4 * 16 <= (size_t)SCR_TOTAL * 4 * sizeof(T)))) {
                error = "ks_rollout: a time budget needs the wave form of the rollout kernel (ks_rollout_plan: KS_PLAN_WAVES)";
                return KS_ERR_STATE;
            }
            const ObsOut<T> out{(T*)ra->sim_obs, (T*)ra->sim_reward, ra->sim_done, (T*)ra->sim_info, (T*)ra->sim_final_obs, cfg.horizon, cfg.auto_reset, cfg.obs_env_major};
            const bool capturing = stream_is_capturing(s);
            ObsOut<T>* slot = pinned_record(capturing, h_out, (unsigned)h_out_next++, captured_out);
            if (!slot) { error = "ks_rollout: more captured calls than the context keeps output records for"; return KS_ERR_STATE; }
            *slot = out;
            out_valid = false;                                     // a following ks_step re-sends its own record
            HIPCHK(hipMemcpyAsync(d_out, slot, sizeof out, hipMemcpyHostToDevice, s));
            ks_rollout_args* rslot = pinned_record(capturing, h_ra, h_ra_next++, captured_ra);
            if (!rslot) { error = "ks_rollout: more captured calls than the context keeps argument records for"; return KS_ERR_STATE; }
            *rslot = *ra;
            HIPCHK(hipMemcpyAsync(d_ra, rslot, sizeof *ra, hipMemcpyHostToDevice, s));
            // more groups than resident workgroups: the ready queue (k_rollout); else one group per workgroup, nothing to deal
            const bool use_queue = plan_queue();
            const bool wave_free = plan_waves() && (size_t)(((ra->h1 + 15) / 16 + (ra->h2 + 15) / 16) * 4) * 4 * 16 <= (size_t)SCR_TOTAL * 4 * sizeof(T);
            if (use_queue) {
                hipLaunchKernelGGL(k_rollout_queue_init, dim3(1), dim3(256), 0, s, d_queue, n_groups);
            }
            if (wave_free && rollout_phase_deal != 0 && n_models == 1 && ra->budget_ticks == 0) {
                if (rollout_phase_deal == 2 && N <= PHASE_ENVS_MAX) hipLaunchKernelGGL(k_slots_by_phase, dim3(1), dim3(PHASE_THREADS), 0, s, (const int64_t*)ra->t, N, n_wg * lpw, b.slot_env);
                else if (lpw <= 16) hipLaunchKernelGGL(k_slots_by_phase_in_groups, dim3((n_groups + 255) / 256), dim3(256), 0, s, (const int64_t*)ra->t, n_groups, lpw, b.slot_env);
            }
#define KS_ROLLOUT_CASE(A, B)                                                                                                                         \
    if ((ra->h1 + 15) / 16 == A && (ra->h2 + 15) / 16 == B) {                                                                                                 \
        HIPCHK(hipFuncSetAttribute((const void*)k_rollout<A, B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_lds));                        \
        hipLaunchKernelGGL((k_rollout<A, B>), dim3(n_groups < resident_wgs ? n_groups : resident_wgs), dim3(WG), step_lds, s, d_model, b, (const Buffers<T>*)d_b, N, cfg.frame_skip,              \
                           cfg.solver_iterations, lpw, (int)cfg.pair_memory, (const ObsOut<T>*)d_out, (const ks_rollout_args*)d_ra, n_iter,                                              \
                           (use_queue || !rollout_round_robin) ? n_groups : -n_groups, use_queue ? d_queue : (int*)nullptr, (int)wave_free);                                                                       \
        HIPCHK(hipGetLastError());                                                                                                                    \
        return KS_OK;                                                                                                                                 \
    }
            KS_ROLLOUT_CASE(16, 16)
            KS_ROLLOUT_CASE(25, 19)
            KS_ROLLOUT_CASE(8, 8)
            KS_ROLLOUT_CASE(4, 4)
#undef KS_ROLLOUT_CASE
            error = "ks_rollout: hidden widths must be 256-256, 400-300, 128-128 or 64-64";
            return KS_ERR_INVALID;
        }
    }
    // more groups than resident workgroups: the ready queue (k_rollout); else one group per workgroup, nothing to deal
    bool plan_queue() const { return rollout_queue && n_groups > resident_wgs && n_groups <= ROLLOUT_QCAP; }
    // one group per workgroup: its four waves run free (k_rollout, round 6); KS_ROLLOUT_WAVES=0 keeps them joined by barriers
    bool plan_waves() const {
        return rollout_waves && !plan_queue() && n_groups <= resident_wgs && lpw == EPW_MAX && LANE_STRIDE == SUBS &&
               (size_t)(wg_rays_words(4, WAVE) + wg_obs_words(4) + 4) <= (size_t)SCR_TOTAL * 4;
    }
    int rollout_plan(int32_t* mode, int32_t* groups, int32_t* workgroups) override {
        if (!model_loaded) { error = "ks_rollout_plan before ks_load_model"; return KS_ERR_STATE; }
        if (mode) *mode = plan_waves() ? KS_PLAN_WAVES : (n_groups <= resident_wgs ? KS_PLAN_WORKGROUPS : (plan_queue() ? KS_PLAN_QUEUE : (rollout_round_robin ? KS_PLAN_ROUND_ROBIN : KS_PLAN_RUNS)));
        if (groups) *groups = n_groups;
        if (workgroups) *workgroups = n_groups < resident_wgs ? n_groups : resident_wgs;
        return KS_OK;
    }
    int substep(const void* ctrl, hipStream_t s) override {
        if (!model_loaded) { error = "ks_substep before ks_load_model"; return KS_ERR_STATE; }
        hipLaunchKernelGGL((k_substep<T, USE_LDS>), dim3(n_wg), dim3(WG), step_lds, s, d_model, b, (const T*)ctrl, cfg.n_envs,
                           cfg.solver_iterations, lpw, cfg.contact_tap);
        HIPCHK(hipGetLastError());
        return KS_OK;
    }
    int get_state(void* qpos, void* qvel, void* warm, void* contact, int32_t* ncon, int32_t* status, hipStream_t s) override {
        const size_t N = cfg.n_envs;
        if (qpos) HIPCHK(hipMemcpyAsync(qpos, b.qpos, NQ * N * sizeof(T), hipMemcpyDefault, s));
        if (qvel) HIPCHK(hipMemcpyAsync(qvel, b.qvel, NV * N * sizeof(T), hipMemcpyDefault, s));
        if (warm) HIPCHK(hipMemcpyAsync(warm, b.warm, NV * N * sizeof(T), hipMemcpyDefault, s));
        if (contact) HIPCHK(hipMemcpyAsync(contact, b.contact, (size_t)NCON_MAX * CON_STRIDE * N * sizeof(T), hipMemcpyDefault, s));
        if (ncon) HIPCHK(hipMemcpyAsync(ncon, b.ncon, N * sizeof(int32_t), hipMemcpyDefault, s));
        if (status) HIPCHK(hipMemcpyAsync(status, b.status, N * sizeof(int32_t), hipMemcpyDefault, s));
        return KS_OK;
    }
    int set_state(const void* qpos, const void* qvel, const void* warm, hipStream_t s) override {
        const size_t N = cfg.n_envs;
        if (qpos) HIPCHK(hipMemcpyAsync(b.qpos, qpos, NQ * N * sizeof(T), hipMemcpyDefault, s));
        if (qvel) HIPCHK(hipMemcpyAsync(b.qvel, qvel, NV * N * sizeof(T), hipMemcpyDefault, s));
        if (warm) HIPCHK(hipMemcpyAsync(b.warm, warm, NV * N * sizeof(T), hipMemcpyDefault, s));
        return KS_OK;
    }
    int set_env_params(const void* mass, const void* mu, hipStream_t s) override {
        if (!model_loaded) { error = "ks_set_env_params before ks_load_model"; return KS_ERR_STATE; }
        const size_t N = cfg.n_envs;
        if (mass) HIPCHK(hipMemcpyAsync(b.envp, mass, N * sizeof(T), hipMemcpyDefault, s));
        if (mu) HIPCHK(hipMemcpyAsync(b.envp + N, mu, N * sizeof(T), hipMemcpyDefault, s));
        return KS_OK;
    }
    int obs_from_snapshot(const void* snap, const void* rays, void* obs, void* reward, uint8_t* done, void* info, hipStream_t s) override {
        if (!model_loaded) { error = "ks_obs_from_snapshot before ks_load_model"; return KS_ERR_STATE; }
        if (!snap || !rays || !obs) { error = "ks_obs_from_snapshot: snapshot, rays and obs are required"; return KS_ERR_INVALID; }
        const size_t N = cfg.n_envs;
        HIPCHK(hipMemcpyAsync(b.snap, snap, (size_t)SNAP_TOTAL * N * sizeof(T), hipMemcpyDefault, s));
        HIPCHK(hipMemcpyAsync(b.rays, rays, (size_t)NRAY * N * sizeof(T), hipMemcpyDefault, s));
        hipLaunchKernelGGL((k_obs<T>), dim3(blocks()), dim3(WAVE), 0, s, d_model, b, (int)N, 2,
                           ObsOut<T>{(T*)obs, (T*)reward, done, (T*)info, nullptr, 0, 0, cfg.obs_env_major});
        HIPCHK(hipGetLastError());
        return KS_OK;
    }
    int kernel_time(int reset, double* avg_ms, int64_t* launches) override {
        HIPCHK(hipDeviceSynchronize());
        for (int i = 0; i < ev_used; i++) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, ev0[i], ev1[i]));
            ev_ms += ms;
            ev_launches++;
        }
        ev_used = 0;
        if (avg_ms) *avg_ms = ev_launches ? ev_ms / (double)ev_launches : 0.0;
        if (launches) *launches = ev_launches;
        if (reset) { ev_ms = 0; ev_launches = 0; step_calls = 0; }
        return KS_OK;
    }
};

}  // namespace

struct ks_ctx {
    CtxBase* impl;
};

extern "C" {

int ks_version(void) { return 1; }

void ks_default_config(ks_config* cfg) {
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->n_envs = 1024;
    cfg->frame_skip = 15;
    cfg->horizon = 30;
    cfg->solver_iterations = 20;   /* early exit on convergence; 6 truncated 2.3 % of the substeps (profiles/r03_solver_cap.txt) */
    cfg->precision = 32;
    cfg->auto_reset = 0;
    cfg->obs_env_major = 1;
    cfg->envs_per_wave = 0;
    cfg->contact_tap = 0;
    cfg->pair_memory = 1;
}

const char* ks_last_error(const ks_ctx* ctx) { return ctx ? ctx->impl->error.c_str() : g_create_error.c_str(); }

int ks_create(const ks_config* cfg, int device, ks_ctx** out) {
    if (!cfg || !out || cfg->n_envs <= 0 || cfg->frame_skip <= 0 || cfg->solver_iterations <= 0 || (cfg->precision != 32 && cfg->precision != 64)) {
        g_create_error = "ks_create: invalid configuration";
        return KS_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        g_create_error = "ks_create: no usable HIP device (this library has no CPU path)";
        return KS_ERR_NO_DEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) { g_create_error = "ks_create: hipSetDevice failed"; return KS_ERR_HIP; }
    CtxBase* impl = cfg->precision == 32 ? (CtxBase*)new Ctx<float>() : (CtxBase*)new Ctx<double>();
    impl->cfg = *cfg;
    impl->device = device;
    int r = cfg->precision == 32 ? static_cast<Ctx<float>*>(impl)->init() : static_cast<Ctx<double>*>(impl)->init();
    if (r != KS_OK) { g_create_error = impl->error; delete impl; return r; }
    *out = new ks_ctx{impl};
    return KS_OK;
}

void ks_destroy(ks_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->impl->device);
    (void)hipDeviceSynchronize();
    delete ctx->impl;
    delete ctx;
}

int ks_load_model(ks_ctx* ctx, const void* blob, size_t n) {
    if (!ctx || !blob) return KS_ERR_INVALID;
    return ctx->impl->load_models(1, &blob, &n);
}
int ks_load_models(ks_ctx* ctx, int32_t n_models, const void* const* blobs, const size_t* nbytes) {
    if (!ctx || !blobs || !nbytes) return KS_ERR_INVALID;
    for (int k = 0; k < n_models; k++)
        if (!blobs[k]) return KS_ERR_INVALID;
    return ctx->impl->load_models(n_models, blobs, nbytes);
}
int ks_reset(ks_ctx* ctx, const int32_t* ids, int32_t n, const void* q0, const void* hq, void* obs, void* stream) {
    if (!ctx) return KS_ERR_INVALID;
    return ctx->impl->reset(ids, n, q0, hq, nullptr, nullptr, obs, (hipStream_t)stream);
}
int ks_reset_objects(ks_ctx* ctx, const int32_t* ids, int32_t n, const void* q0, const void* hq, const int32_t* object_id, const void* mass_friction,
                     void* obs, void* stream) {
    if (!ctx) return KS_ERR_INVALID;
    return ctx->impl->reset(ids, n, q0, hq, object_id, mass_friction, obs, (hipStream_t)stream);
}
int ks_step(ks_ctx* ctx, const void* action, void* obs, void* reward, uint8_t* done, void* info, void* final_obs, void* stream) {
    if (!ctx) return KS_ERR_INVALID;
    return ctx->impl->step(action, obs, reward, done, info, final_obs, (hipStream_t)stream);
}
int ks_get_state(ks_ctx* ctx, void* qpos, void* qvel, void* warm, void* contact, int32_t* ncon, int32_t* status, void* stream) {
    if (!ctx) return KS_ERR_INVALID;
    return ctx->impl->get_state(qpos, qvel, warm, contact, ncon, status, (hipStream_t)stream);
}
int ks_set_state(ks_ctx* ctx, const void* qpos, const void* qvel, const void* warm, void* stream) {
    if (!ctx) return KS_ERR_INVALID;
    return ctx->impl->set_state(qpos, qvel, warm, (hipStream_t)stream);
}
int ks_set_env_params(ks_ctx* ctx, const void* obj_mass, const void* obj_mu, void* stream) {
    if (!ctx) return KS_ERR_INVALID;
    return ctx->impl->set_env_params(obj_mass, obj_mu, (hipStream_t)stream);
}
int ks_rollout(ks_ctx* ctx, int32_t n_iter, const ks_rollout_args* args_host, void* stream) {
    if (!ctx) return KS_ERR_INVALID;
    return ctx->impl->rollout(n_iter, args_host, (hipStream_t)stream);
}
int ks_rollout_plan(ks_ctx* ctx, int32_t* mode, int32_t* groups, int32_t* workgroups) {
    if (!ctx) return KS_ERR_INVALID;
    return ctx->impl->rollout_plan(mode, groups, workgroups);
}
int ks_substep(ks_ctx* ctx, const void* ctrl, void* stream) {
    if (!ctx) return KS_ERR_INVALID;
    return ctx->impl->substep(ctrl, (hipStream_t)stream);
}
int ks_obs_from_snapshot(ks_ctx* ctx, const void* snap, const void* rays, void* obs, void* reward, uint8_t* done, void* info, void* stream) {
    if (!ctx) return KS_ERR_INVALID;
    return ctx->impl->obs_from_snapshot(snap, rays, obs, reward, done, info, (hipStream_t)stream);
}
int ks_kernel_time(ks_ctx* ctx, int reset, double* avg_ms, int64_t* launches) {
    if (!ctx) return KS_ERR_INVALID;
    return ctx->impl->kernel_time(reset, avg_ms, launches);
}

}  // extern "C"
