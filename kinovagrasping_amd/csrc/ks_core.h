// ks_core.h -- one-env-per-lane mj_step for the Kinova j2s7s300 end effector (S1-S7).
//
// What it replaces: `self._sim.step()` at kinova_gripper_env.py:1535 (MuJoCo 1.50 mj_step through
// mujoco-py), 15 times per env.step() (frame_skip, kinova_gripper_env.py:51,1516).
//
// Design (MI355X-first, NOT a translation of a generic engine):
//  * one env per wavefront lane; fixed topology (3 slides, 3 planar two-link fingers, one free
//    object) is exploited analytically: block-structured mass matrix, closed-form finger
//    Coriolis terms, contact Jacobians rebuilt from contact geometry instead of being stored;
//  * per-lane dynamic state that needs runtime indexing (body poses, contact list, per-contact
//    solver scalars) lives in a lane-interleaved scratch `S` (LDS, stride 64 floats -> bank ==
//    lane, conflict free); everything with static indexing stays in registers;
//  * hull vertex / face tables are wave-uniform reads (scalar loads through the constant cache);
//  * constraint solver: exact Newton on the 15-dim primal problem (MuJoCo's default solver for
//    this model) with a fixed iteration count so lanes stay in lock step.
//
// The code is KS_HD so the same source is lane-checked on the CPU against the fp64 oracle.
#pragma once
#include <type_traits>
#include <utility>

#include "ks_model.h"

namespace ks {

// Which stages are real (out-of-line) device functions.  The narrow-phase queries are inlined into `collision`
// (measured: +2.3 % env-steps/s, and the parity suite is indifferent to it now - see DESIGN.md on the round-1 "inlining
// breaks parity" symptom); -DKS_OUTLINE_NARROW restores the calls.  Of the other stages the solver is inlined into the
// substep as well (+1.9 %; collision and dynamics measured +1 % alone and nothing in combination, so they stay calls):
// KS_INLINE_COLLISION / KS_INLINE_DYNAMICS / KS_OUTLINE_SOLVER are the experiment switches.
#ifdef KS_OUTLINE_NARROW
#define KS_NARROW KS_FN
#else
#define KS_NARROW KS_HD
#endif
#ifdef KS_INLINE_COLLISION
#define KS_FN_COLLISION KS_HD
#else
#define KS_FN_COLLISION KS_FN
#endif
#ifdef KS_INLINE_DYNAMICS
#define KS_FN_DYNAMICS KS_HD
#else
#define KS_FN_DYNAMICS KS_FN
#endif
#ifdef KS_OUTLINE_SOLVER
#define KS_FN_SOLVER KS_FN
#else
#define KS_FN_SOLVER KS_HD
#endif

// ---------------------------------------------------------------- scratch layout (units of T)
// body poses b = 2..9: 12 each (R row-major 9, p 3)
constexpr int SCR_BP = 0;
constexpr int SCR_AX = SCR_BP + 8 * 12;      // world slide axes, 3 x 3 (+3 pad: every region starts on a 16-byte boundary)
// per-env randomised parameters (BASELINE config 5): object mass, object-hand friction - in the padding
constexpr int SCR_ENVP = SCR_AX + 9;
constexpr int SCR_CON = SCR_AX + 12;
constexpr int CON_STRIDE = 20;
// per contact: 0-2 pos, 3-5 normal, 6 dist, 7 mu, 8 bodies + pair (b1 + 16*b2 + 256*pair index), 9 R, 10-13 aref[4],
//              14-16 J.a basis (n,t1,t2), 17-19 J.p basis
// collision staging: contacts are detected pair by pair into per-pair slots of (pos3, normal3, dist, mu,
// bodies) records - 4 slots for a plane pair, 1 for a hull pair, assigned in pair order - and then merged
// in pair order with the per-pair counts in SCR_PC.
// The multi-geom build (16 plane pairs x 4 + 78 hull pairs = 142 static slots = 5 KB per env) hands the slots out per substep to the pairs
// that passed the culls instead (DYNAMIC_SLOTS: a pair's slot of this substep is kept in SCR_SLOT), so that 16 env blocks still fit a
// compute unit's LDS beside the 96 pair records.
constexpr bool DYNAMIC_SLOTS = MULTI_GEOM;
#ifndef KS_MG_NSTAGE
#define KS_MG_NSTAGE 80         // (>= 4 x live plane pairs + live hull pairs of a substep - a bowl pressed by the hand: ~40 -; the rest of the region is what the solver's
#endif                          //  contact-basis cache gets: 80 records = 720 words = 16 bases of 45)
constexpr int NSTAGE = MULTI_GEOM ? KS_MG_NSTAGE : 80, STAGE_REC = 9, STAGE_WORDS = NSTAGE * STAGE_REC;   // 80 x 9 = 720 = 15 cached bases
constexpr int SCR_PC = SCR_CON + NCON_MAX * CON_STRIDE;
constexpr int SCR_SLOT = SCR_PC + NPAIR_MAX;                                   // [NPAIR_MAX] (dynamic slots only)
constexpr int SCR_STAGE = SCR_SLOT + (DYNAMIC_SLOTS ? NPAIR_MAX : 0);
// smooth dynamics of the substep, written by the role lanes (fingers, object, slides) and read by row:
// hand mass matrix 9x9, object mass matrix 6x6, qfrc_smooth (slide entries without the finger links' bias),
// per-finger bias on the three slides
constexpr int SCR_MH = SCR_STAGE + STAGE_WORDS;
static_assert(NSTAGE * STAGE_REC <= STAGE_WORDS, "staging records fit");
constexpr int SCR_MO = SCR_MH + 81;
constexpr int SCR_QF = SCR_MO + 36;
constexpr int SCR_SB = SCR_QF + NV;
constexpr int SCR_GP = ((SCR_SB + 9 + 3) / 4) * 4;   // world poses of geoms 1..8 (R row-major 9, p 3), refreshed every substep
// the env's state and per-step constants (GPU: the stepping kernel keeps them here instead of in per-lane
// registers / stack, the stages reach them through generic pointers): LaneState (qpos 16, qvel 15, warm 15),
// hand rotation 9, controls 9
constexpr int SCR_STATE = SCR_GP + (NGEOM - 1) * 12;
constexpr int SCR_TOTAL = SCR_STATE + 64;
// The staging region is dead once the contacts are merged: the solver reuses them as a cache of
// the contact basis Jacobians (45 values per contact) so that they are built once per substep, not 2x per
// Newton iteration; contacts that do not fit are rebuilt on the fly.
constexpr int SCR_BCACHE = SCR_STAGE;
#ifndef KS_BC_STRIDE
#ifdef KS_MULTI_GEOM
#define KS_BC_STRIDE 45         // unpadded: 16 cached bases instead of 15 - a round bowl rests on 16 contacts, and a contact beyond the cache is rebuilt twice per Newton
#else                           // iteration (measured, 4096 envs: BowlS 3.64 -> 2.85 ms per env-step, BowlB 5.4 -> 4.7, RBowlS 2.02 -> 1.83; bottles and cubes unchanged)
#define KS_BC_STRIDE 48
#endif
#endif
constexpr int BC_STRIDE = KS_BC_STRIDE;                // 3 x 15 values (standard build: padded to whole 16-byte vectors)
constexpr int NBCACHE = (SCR_MH - SCR_STAGE) / BC_STRIDE;
static_assert(SCR_CON % 4 == 0 && SCR_STAGE % 4 == 0 && SCR_TOTAL % 4 == 0 && CON_STRIDE % 4 == 0, "16-byte aligned scratch regions");

// A team = the SUBS lanes that work on one env (SUBS = 16 on the GPU: one DPP row; 1 on the host).  The
// lanes keep identical copies of the env state; they share the plane-hull vertex scans and split the
// per-pair / per-contact loops.  Teams are aligned groups of lanes of one wave.
template <int SUBS> struct Team {
    int sub;
#if defined(__HIP_DEVICE_COMPILE__)
    // x + (x of the lane this one is paired with under the DPP control word): one v_add_f32_dpp
    template <int CTRL> static __device__ __forceinline__ float dpp_add(float x) {
        return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
    }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    // x of the lane selected by the DPP control word (32-bit types)
    template <int CTRL, typename T> static __device__ __forceinline__ T dpp_get(T x) {
        return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
    }
#endif
    template <typename T> KS_HD T sum(T x) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (sizeof(T) == 4) {
            // row_ror:8, row_ror:4 fold the four quads of the row; quad_perm [1,0,3,2], [2,3,0,1] the quad
            if constexpr (SUBS >= 16) x = dpp_add<0x128>(x);
            if constexpr (SUBS >= 8) x = dpp_add<0x124>(x);
            if constexpr (SUBS >= 4) x = dpp_add<0x4E>(x);
            if constexpr (SUBS >= 2) x = dpp_add<0xB1>(x);
            static_assert(SUBS == 1 || SUBS == 4 || SUBS == 16, "team sizes: 1, 4 or 16 lanes");
        } else {
            if constexpr (SUBS >= 16) x += __shfl_xor(x, 8);
            if constexpr (SUBS >= 8) x += __shfl_xor(x, 4);
            if constexpr (SUBS >= 4) x += __shfl_xor(x, 2);
            if constexpr (SUBS >= 2) x += __shfl_xor(x, 1);
        }
#endif
        return x;
    }
    // x of team lane SRC (a compile-time lane: one DPP row_newbcast mov)
    template <int SRC, typename T> KS_HD T bcast(T x) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (SUBS == 1) return x;
        else if constexpr (sizeof(T) == 4) {
            static_assert(SUBS == 16, "row broadcast needs a full DPP row");
            return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x150 + SRC, 0xf, 0xf, false));
        } else return __shfl(x, SRC, SUBS);
#else
        return x;
#endif
    }
    // smallest d over the team, lowest index among equal values; every lane gets the result
    template <typename T> KS_HD void argmin(T& d, int& i) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (sizeof(T) == 4 && SUBS == 16) {
            // two DPP butterflies: the minimum, then the lowest index among the lanes that hold it
            T dm = d;
            dm = kmin(dm, dpp_get<0x128>(dm)); dm = kmin(dm, dpp_get<0x124>(dm));
            dm = kmin(dm, dpp_get<0x4E>(dm)); dm = kmin(dm, dpp_get<0xB1>(dm));
            int c = (d == dm) ? i : 0x7fffffff;
            c = kmin(c, dpp_get<0x128>(c)); c = kmin(c, dpp_get<0x124>(c));
            c = kmin(c, dpp_get<0x4E>(c)); c = kmin(c, dpp_get<0xB1>(c));
            d = dm; i = c;
        } else {
            KS_UNROLL
            for (int mask = SUBS / 2; mask >= 1; mask >>= 1) {
                const T od = __shfl_xor(d, mask);
                const int oi = __shfl_xor(i, mask);
                if (od < d || (od == d && oi < i)) { d = od; i = oi; }
            }
        }
#endif
    }
    // exclusive prefix sum of x over the team's lanes (lane order), team total in `total`
    KS_HD int scan(int x, int& total) const {
        int incl = x;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (SUBS == 16) {
            // row_shr:1,2,4,8 with zero fill
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);
            total = __builtin_amdgcn_update_dpp(0, incl, 0x15F, 0xf, 0xf, false);
        } else {
            KS_UNROLL
            for (int d = 1; d < SUBS; d <<= 1) {
                const int o = __shfl_up(incl, d, SUBS);
                if (sub >= d) incl += o;
            }
            total = __shfl(incl, SUBS - 1, SUBS);
        }
#else
        total = incl;
#endif
        return incl - x;
    }
    // bit k set: team lane k voted true
    KS_HD unsigned ballot(bool pred) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (SUBS == 1) return pred ? 1u : 0u;
        else {
            const unsigned long long b = __ballot(pred);
            const int base = (int)(__lane_id() & ~(SUBS - 1));
            return (unsigned)(b >> base) & ((1u << SUBS) - 1u);
        }
#else
        return pred ? 1u : 0u;
#endif
    }
    // LDS writes of the team members become visible to each other (one wave: program order + a fence)
    KS_HD void sync() const {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (SUBS > 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
#endif
    }
};

// Lane-interleaved scratch accessor: element k of this lane lives at base[k*stride].  P is the pointer
// type: an address_space(3) pointer on the GPU (real ds_read / ds_write instead of flat accesses through
// a generic pointer), a plain pointer for the fp64 / host instantiations.
template <typename T, typename P = T*> struct Scratch {
    P base;
    int stride;
    struct Ref {
        P p;
        KS_HD operator T() const { return *p; }
        KS_HD const Ref& operator=(T v) const { *p = v; return *this; }
        KS_HD const Ref& operator=(const Ref& o) const { *p = T(*o.p); return *this; }   // element copy, not proxy copy
    };
    KS_HD Ref operator()(int k) const { return Ref{base + k * stride}; }
};

// Contiguous scratch accessor: element k of this env lives at base[k] (the env's private block).  Used for the
// LDS scratch of the stepping kernels: the 16 lanes of a team read the same addresses (LDS broadcast), adjacent
// elements merge into ds_read_b64 / b128, and with a block size of 12 (mod 32) words the four teams of a wave
// hit disjoint bank groups.
template <typename T, typename P = T*> struct ScratchC {
    P base;
    struct Ref {
        P p;
        KS_HD operator T() const { return *p; }
        KS_HD const Ref& operator=(T v) const { *p = v; return *this; }
        KS_HD const Ref& operator=(const Ref& o) const { *p = T(*o.p); return *this; }
    };
    KS_HD Ref operator()(int k) const { return Ref{base + k}; }
};

// Convex-hull vertex tables as the collision code sees them: on the GPU they are staged in LDS once per
// launch (wave-uniform ds_read broadcasts, no scalar-cache thrash, no per-lane 64-bit address math).
// Everything the narrow phase needs to know about one collision pair, gathered once per launch into one 96-byte
// LDS record: a pair costs one burst of ds_read_b128 instead of a chain of dependent model lookups.
template <typename T> struct alignas(16) PairRec {
    int g1, g2;
    T margin, mu;
    T rbound1, size1[3];
    T rbound2, size2[3];
    int body1, body2;
    int n1, n2;                // padded hull vertex counts; n2 of a plane pair is the true count
    int slot, obj_hand;        // first staging record of the pair (4 per plane pair, 1 per hull pair, pair order);
                               // obj_hand: bit 0 = object vs hand geom (the pair whose friction may be set per env),
                               // bits 4-7 / 8-11 = mesh ids of the two geoms, bits 12-18 = the pair's index
    KS_TAB const T* V1; KS_TAB const T* V2;
    KS_TAB const unsigned short* off1; KS_TAB const unsigned short* adj1;
    KS_TAB const unsigned short* off2; KS_TAB const unsigned short* adj2;
};
constexpr int PAIR_INDEX_MASK = 127;

template <typename T> struct Hulls {
    KS_TAB const T* vert[NMESH];   // [nvert_pad][4]
    int nvert[NMESH], nvert_pad[NMESH];
    KS_TAB const unsigned short* adj_off[NMESH];   // hull graph, 4-neighbour chunks (hill-climbing support queries)
    KS_TAB const unsigned short* adj[NMESH];
    int npair, nhull;
    KS_LDS const PairRec<T>* pair;             // [npair]
    unsigned char hull_pi[NPAIR_MAX];          // pair index of the k-th hull-hull pair
    int nplane;
    unsigned char plane_pi[NPAIR_MAX];         // pair index of the k-th plane pair
};
// pair pi is ground plane vs hull
template <typename T> KS_HD bool is_plane_pair(const Model<T>& m, int pi) { return m.pair_g1[pi] == 0; }
template <typename T> KS_HD int hull_pair_count(const Model<T>& m) {
    int nh = 0;
    for (int pi = 0; pi < m.npair; pi++) nh += is_plane_pair(m, pi) ? 0 : 1;
    return nh;
}
// pair bookkeeping of the descriptor (counts, plane / hull pair lists as the model holds them); `pair` is set by the caller
template <typename T> KS_HD void hulls_set_pairs(const Model<T>& m, Hulls<T>& hu) {
    hu.npair = m.npair;
    hu.nhull = m.nhull;
    hu.nplane = m.nplane;
    static_assert(NPAIR_MAX % 4 == 0, "pair lists copied as words");
    const unsigned* sp = (const unsigned*)m.plane_order;
    const unsigned* sh = (const unsigned*)m.hull_order;
    unsigned* dp = (unsigned*)hu.plane_pi;
    unsigned* dh = (unsigned*)hu.hull_pi;
    KS_UNROLL
    for (int k = 0; k < NPAIR_MAX / 4; k++) { dp[k] = sp[k]; dh[k] = sh[k]; }
}
template <typename T> KS_HD void fill_pair_rec(const Model<T>& m, const Hulls<T>& hu, int pi, PairRec<T>& r) {
    const int g1 = m.pair_g1[pi], g2 = m.pair_g2[pi];
    r.g1 = g1; r.g2 = g2; r.margin = m.pair_margin[pi]; r.mu = m.pair_mu[pi];
    int slot = 0;
    for (int j = 0; j < pi; j++) slot += is_plane_pair(m, j) ? 4 : 1;
    r.slot = slot;
    r.obj_hand = ((g1 != 0 && g2 >= OBJ_GEOM) ? 1 : 0) | ((g1 != 0 ? m.geom_mesh[g1] : 0) << 4) | (m.geom_mesh[g2] << 8) | (pi << 12);
    r.rbound1 = m.geom_rbound[g1]; r.rbound2 = m.geom_rbound[g2];
    for (int k = 0; k < 3; k++) { r.size1[k] = m.geom_size[g1][k]; r.size2[k] = m.geom_size[g2][k]; }
    r.body1 = m.geom_body[g1]; r.body2 = m.geom_body[g2];
    const int mesh2 = m.geom_mesh[g2];
    r.V2 = hu.vert[mesh2]; r.off2 = hu.adj_off[mesh2]; r.adj2 = hu.adj[mesh2];
    if (g1 == 0) {
        r.n1 = 0; r.n2 = hu.nvert[mesh2];
        r.V1 = r.V2; r.off1 = r.off2; r.adj1 = r.adj2;
    } else {
        const int mesh1 = m.geom_mesh[g1];
        r.n1 = hu.nvert_pad[mesh1]; r.n2 = hu.nvert_pad[mesh2];
        r.V1 = hu.vert[mesh1]; r.off1 = hu.adj_off[mesh1]; r.adj1 = hu.adj[mesh1];
    }
}

template <typename T> struct LaneState {
    T qpos[NQ], qvel[NV], warm[NV];
};

// kinematics of the hand + object kept in registers for the smooth dynamics
template <typename T> struct Kin {
    T ax[3][3];              // world slide axes
    T p7[3];
    T Rp[3][9], pp[3][3];    // proximal links
    T Rd[3][9], pd[3][3];    // distal links
    T Ro[9], po[3];
};

// ---------------------------------------------------------------- S1 forward kinematics
template <typename T, typename S>
KS_HD void forward_kinematics(const Model<T>& m, const T* qpos, const T* R7, Kin<T>& k, S scr, bool write_poses = true) {
    KS_UNROLL
    for (int i = 0; i < 3; i++) mulRv(k.ax[i], R7, m.slide_axis[i]);
    KS_UNROLL
    for (int i = 0; i < 3; i++) k.p7[i] = m.l7_pos[i] + k.ax[0][i] * qpos[0] + k.ax[1][i] * qpos[1] + k.ax[2][i] * qpos[2];
    KS_UNROLL
    for (int f = 0; f < 3; f++) {
        T Rb[9], Rz[9], t[3];
        mulRR(Rb, R7, m.fbase_R[f]);
        T a = qpos[3 + 2 * f], c = kcos(a), s = ksin(a);
        Rz[0] = c; Rz[1] = -s; Rz[2] = 0; Rz[3] = s; Rz[4] = c; Rz[5] = 0; Rz[6] = 0; Rz[7] = 0; Rz[8] = 1;
        mulRR(k.Rp[f], Rb, Rz);
        mulRv(t, R7, m.fbase_pos[f]);
        add3(k.pp[f], k.p7, t);
        mulRR(Rb, k.Rp[f], m.ftip_R[f]);
        a = qpos[4 + 2 * f]; c = kcos(a); s = ksin(a);
        Rz[0] = c; Rz[1] = -s; Rz[3] = s; Rz[4] = c;
        mulRR(k.Rd[f], Rb, Rz);
        mulRv(t, k.Rp[f], m.ftip_pos[f]);
        add3(k.pd[f], k.pp[f], t);
    }
    T q[4] = {qpos[12], qpos[13], qpos[14], qpos[15]};
    quatnormalize(q);
    quat2mat(k.Ro, q);
    copy3(k.po, &qpos[9]);
    // body poses -> scratch (runtime-indexed by collision / contact code)
    if (!write_poses) return;
    KS_UNROLL
    for (int j = 0; j < 9; j++) scr(SCR_AX + j) = k.ax[j / 3][j % 3];
    KS_UNROLL
    for (int j = 0; j < 9; j++) scr(SCR_BP + j) = R7[j];
    KS_UNROLL
    for (int j = 0; j < 3; j++) scr(SCR_BP + 9 + j) = k.p7[j];
    KS_UNROLL
    for (int f = 0; f < 3; f++) {
        int o1 = SCR_BP + (1 + 2 * f) * 12, o2 = SCR_BP + (2 + 2 * f) * 12;
        KS_UNROLL
        for (int j = 0; j < 9; j++) { scr(o1 + j) = k.Rp[f][j]; scr(o2 + j) = k.Rd[f][j]; }
        KS_UNROLL
        for (int j = 0; j < 3; j++) { scr(o1 + 9 + j) = k.pp[f][j]; scr(o2 + 9 + j) = k.pd[f][j]; }
    }
    KS_UNROLL
    for (int j = 0; j < 9; j++) scr(SCR_BP + 7 * 12 + j) = k.Ro[j];
    KS_UNROLL
    for (int j = 0; j < 3; j++) scr(SCR_BP + 7 * 12 + 9 + j) = k.po[j];
}

// world pose of geom g (1..8) from the body pose in scratch
template <typename T, typename S> KS_HD void geom_pose(const Model<T>& m, S scr, int g, T* R, T* p) {
    int o = SCR_BP + (m.geom_body[g] - 2) * 12;
    T Rb[9], pb[3], t[3];
    KS_UNROLL
    for (int j = 0; j < 9; j++) Rb[j] = scr(o + j);
    KS_UNROLL
    for (int j = 0; j < 3; j++) pb[j] = scr(o + 9 + j);
    mulRR(R, Rb, m.geom_R[g]);
    mulRv(t, Rb, m.geom_pos[g]);
    add3(p, pb, t);
}

// nominal values of the per-env parameters: the model's object mass and the friction of its object-hand pairs
template <typename T> KS_HD void nominal_env_params(const Model<T>& m, T& mass, T& mu) {
    mass = m.mass[NBODY - 1];
    mu = T(1);
    for (int pi = 0; pi < m.npair; pi++)
        if (m.pair_g1[pi] != 0 && m.pair_g2[pi] >= OBJ_GEOM) { mu = m.pair_mu[pi]; break; }
}

// world pose of geom g (1..8) as stored by dynamics_rows
template <typename T, typename S> KS_HD void geom_pose_cached(S scr, int g, T* R, T* p) {
    const int o = SCR_GP + (g - 1) * 12;
    KS_UNROLL
    for (int j = 0; j < 9; j++) R[j] = scr(o + j);
    KS_UNROLL
    for (int j = 0; j < 3; j++) p[j] = scr(o + 9 + j);
}

// ---------------------------------------------------------------- S1-S3 by role lanes
// Forward kinematics + smooth dynamics of one substep, split over the team by ROLE: lanes 0-2 one finger each
// (proximal + distal link), lane 3 the object, lane 4 the slides / link_7.  Every role writes its body poses
// (SCR_BP / SCR_AX) and its part of the mass matrix and of qfrc_smooth = passive - bias + actuator into the
// env's LDS block; consumers read them by row (load_dynamics_row).  With SUBS = 1 one lane plays all roles.
template <typename T, typename S, int SUBS>
KS_FN_DYNAMICS void dynamics_rows(const Model<T>& m, const T* qpos, const T* qvel, const T* ctrl, const T* R7, S scr, Team<SUBS> team) {
    // entries no role writes (finger-finger cross terms, hand-object) stay zero
    for (int k = team.sub; k < 81 + 36; k += SUBS) scr(SCR_MH + k) = T(0);
    team.sync();
    T ax[3][3], p7[3];
    KS_UNROLL
    for (int i = 0; i < 3; i++) mulRv(ax[i], R7, m.slide_axis[i]);
    KS_UNROLL
    for (int i = 0; i < 3; i++) p7[i] = m.l7_pos[i] + ax[0][i] * qpos[0] + ax[1][i] * qpos[1] + ax[2][i] * qpos[2];
    const T g = -m.gravity_z;   // a_com - gravity = a_com + (0,0,g)
    for (int role = team.sub; role < 5; role += SUBS) {
        if (role < 3) {
            const int f = role, hp = 3 + 2 * f, hd = 4 + 2 * f, bP = hp, bD = hd;
            T Rp[9], pp[3], Rd[9], pd[3];
            {
                T Rb[9], Rz[9], t[3];
                mulRR(Rb, R7, m.fbase_R[f]);
                T a = qpos[hp], c = kcos(a), sn = ksin(a);
                Rz[0] = c; Rz[1] = -sn; Rz[2] = 0; Rz[3] = sn; Rz[4] = c; Rz[5] = 0; Rz[6] = 0; Rz[7] = 0; Rz[8] = 1;
                mulRR(Rp, Rb, Rz);
                mulRv(t, R7, m.fbase_pos[f]);
                add3(pp, p7, t);
                mulRR(Rb, Rp, m.ftip_R[f]);
                a = qpos[hd]; c = kcos(a); sn = ksin(a);
                Rz[0] = c; Rz[1] = -sn; Rz[3] = sn; Rz[4] = c;
                mulRR(Rd, Rb, Rz);
                mulRv(t, Rp, m.ftip_pos[f]);
                add3(pd, pp, t);
            }
            const int o1 = SCR_BP + (1 + 2 * f) * 12, o2 = o1 + 12;
            KS_UNROLL
            for (int j = 0; j < 9; j++) { scr(o1 + j) = Rp[j]; scr(o2 + j) = Rd[j]; }
            KS_UNROLL
            for (int j = 0; j < 3; j++) { scr(o1 + 9 + j) = pp[j]; scr(o2 + 9 + j) = pd[j]; }
            const T mP = m.mass[bP], mD = m.mass[bD];
            const T z[3] = {Rp[2], Rp[5], Rp[8]};
            T rPP[3], rDD[3], rDP[3], t[3];
            mulRv(rPP, Rp, m.ipos[bP]);
            mulRv(rDD, Rd, m.ipos[bD]);
            sub3(t, pd, pp);
            add3(rDP, t, rDD);
            T jP[3], jDp[3], jDd[3];
            cross3(jP, z, rPP);
            cross3(jDp, z, rDP);
            cross3(jDd, z, rDD);
            scr(SCR_MH + hp * 9 + hp) = mP * dot3(jP, jP) + mD * dot3(jDp, jDp) + m.izz[bP] + m.izz[bD] + m.armature[hp];
            const T mpd = mD * dot3(jDp, jDd) + m.izz[bD];
            scr(SCR_MH + hp * 9 + hd) = mpd;
            scr(SCR_MH + hd * 9 + hp) = mpd;
            scr(SCR_MH + hd * 9 + hd) = mD * dot3(jDd, jDd) + m.izz[bD] + m.armature[hd];
            const T sP[3] = {mP * jP[0] + mD * jDp[0], mP * jP[1] + mD * jDp[1], mP * jP[2] + mD * jDp[2]};
            KS_UNROLL
            for (int i = 0; i < 3; i++) {
                const T cp = dot3(ax[i], sP), cd = mD * dot3(ax[i], jDd);
                scr(SCR_MH + i * 9 + hp) = cp; scr(SCR_MH + hp * 9 + i) = cp;
                scr(SCR_MH + i * 9 + hd) = cd; scr(SCR_MH + hd * 9 + i) = cd;
            }
            // velocity-product accelerations of the two COMs (planar chain about z): a = w x (w x r)
            const T qvp = qvel[hp], qvd = qvel[hd];
            const T wP = qvp, wD = qvp + qvd;
            T aP[3], aD[3], u[3], zs[3];
            scl3(zs, z, wP);
            cross3(u, zs, rPP); cross3(aP, zs, u);
            cross3(u, zs, t); cross3(aD, zs, u);           // distal origin
            scl3(zs, z, wD);
            cross3(u, zs, rDD); cross3(u, zs, u);
            add3(aD, aD, u);
            aP[2] += g; aD[2] += g;
            T FP[3], FD[3];
            scl3(FP, aP, mP);
            scl3(FD, aD, mD);
            KS_UNROLL
            for (int i = 0; i < 3; i++) scr(SCR_SB + 3 * f + i) = dot3(ax[i], FP) + dot3(ax[i], FD);
            const T bias_p = dot3(jP, FP) + dot3(jDp, FD), bias_d = dot3(jDd, FD);
            scr(SCR_QF + hp) = -m.damping[hp] * qvp - bias_p + m.act[3] * (clampT(ctrl[6 + f], -m.act[4], m.act[4]) - qvp);
            scr(SCR_QF + hd) = -m.damping[hd] * qvd - bias_d;
        } else if (role == 3) {
            // object (free joint: linear world, angular body frame)
            T q[4] = {qpos[12], qpos[13], qpos[14], qpos[15]}, Ro[9];
            quatnormalize(q);
            quat2mat(Ro, q);
            KS_UNROLL
            for (int j = 0; j < 9; j++) scr(SCR_BP + 7 * 12 + j) = Ro[j];
            KS_UNROLL
            for (int j = 0; j < 3; j++) scr(SCR_BP + 7 * 12 + 9 + j) = qpos[9 + j];
            const T mo = scr(SCR_ENVP), ms = mo / m.mass[9];   // per-env mass; the inertia scales with it
            T c[3];
            mulRv(c, Ro, m.ipos[9]);
            KS_UNROLL
            for (int i = 0; i < 3; i++) scr(SCR_MO + i * 6 + i) = mo + m.armature[9 + i];
            const T cb[3] = {m.ipos[9][0], m.ipos[9][1], m.ipos[9][2]};
            const T cc = dot3(cb, cb);
            KS_UNROLL
            for (int a = 0; a < 3; a++) {
                T e[3] = {T(a == 0), T(a == 1), T(a == 2)}, ec[3], w[3];
                cross3(ec, e, cb);
                mulRv(w, Ro, ec);                        // R_a x c (world)
                KS_UNROLL
                for (int i = 0; i < 3; i++) { scr(SCR_MO + i * 6 + 3 + a) = mo * w[i]; scr(SCR_MO + (3 + a) * 6 + i) = mo * w[i]; }
                KS_UNROLL
                for (int b = 0; b < 3; b++)
                    scr(SCR_MO + (3 + a) * 6 + 3 + b) = ms * m.obj_Ib[a * 3 + b] + mo * ((a == b ? cc : T(0)) - cb[a] * cb[b]) + (a == b ? m.armature[12 + a] : T(0));
            }
            // bias: F = m (w x (w x c) + g e_z), T = w x I w ; angular rows in the body frame
            T wl[3] = {qvel[12], qvel[13], qvel[14]}, w[3], u[3], ac[3];
            mulRv(w, Ro, wl);
            cross3(u, w, c); cross3(ac, w, u);
            ac[2] += g;
            T F[3];
            scl3(F, ac, mo);
            T Iwl[3], tl[3];
            mulRv(Iwl, m.obj_Ib, wl);
            scl3(Iwl, Iwl, ms);
            cross3(tl, wl, Iwl);
            T cF[3], cFl[3];
            cross3(cF, c, F);
            mulRtv(cFl, Ro, cF);
            KS_UNROLL
            for (int i = 0; i < 3; i++) {
                scr(SCR_QF + 9 + i) = -m.damping[9 + i] * qvel[9 + i] - F[i];
                scr(SCR_QF + 12 + i) = -m.damping[12 + i] * qvel[12 + i] - (cFl[i] + tl[i]);
            }
        } else {
            // slides: link_7 pose, world slide axes, 3x3 slide block, slide forces without the finger links' bias
            KS_UNROLL
            for (int j = 0; j < 9; j++) { scr(SCR_AX + j) = ax[j / 3][j % 3]; scr(SCR_BP + j) = R7[j]; }
            KS_UNROLL
            for (int j = 0; j < 3; j++) scr(SCR_BP + 9 + j) = p7[j];
            T mt = m.mass[2];
            KS_UNROLL
            for (int b = 3; b <= 8; b++) mt += m.mass[b];
            KS_UNROLL
            for (int i = 0; i < 3; i++) {
                KS_UNROLL
                for (int j = 0; j < 3; j++) scr(SCR_MH + i * 9 + j) = mt * dot3(ax[i], ax[j]) + (i == j ? m.armature[i] : T(0));
                scr(SCR_QF + i) = -m.damping[i] * qvel[i] - m.mass[2] * g * ax[i][2] +
                                  m.act[0] * (clampT(ctrl[2 * i], -m.act[2], m.act[2]) - qvel[i]) + m.act[1] * ctrl[2 * i + 1];
            }
        }
    }
    team.sync();
    // world poses of the collision geoms, one geom per lane
    for (int g = 1 + team.sub; g < m.ngeom; g += SUBS) {
        T R[9], p[3];
        geom_pose(m, scr, g, R, p);
        const int o = SCR_GP + (g - 1) * 12;
        KS_UNROLL
        for (int j = 0; j < 9; j++) scr(o + j) = R[j];
        KS_UNROLL
        for (int j = 0; j < 3; j++) scr(o + 9 + j) = p[j];
    }
    team.sync();
}

// row i of the block-diagonal mass matrix and entry i of qfrc_smooth from the env's LDS block (i >= NV: zeros)
template <typename T, typename S>
KS_HD void load_dynamics_row(S scr, int i, T (&Mrow)[NV], T& qs) {
    KS_UNROLL
    for (int j = 0; j < NV; j++) Mrow[j] = 0;
    qs = 0;
    if (i < 9) {
        KS_UNROLL
        for (int j = 0; j < 9; j++) Mrow[j] = scr(SCR_MH + i * 9 + j);
        qs = scr(SCR_QF + i);
        if (i < 3) qs -= T(scr(SCR_SB + i)) + T(scr(SCR_SB + 3 + i)) + T(scr(SCR_SB + 6 + i));
    } else if (i < NV) {
        KS_UNROLL
        for (int j = 0; j < 6; j++) Mrow[9 + j] = scr(SCR_MO + (i - 9) * 6 + j);
        qs = scr(SCR_QF + i);
    }
}

// ---------------------------------------------------------------- S4 collision
template <typename T> struct Supp {
    T v[3], v1[3], v2[3];
    int i1, i2;                // hull vertex ids of v1 / v2 (support points only)
};

// (TV: the hull tables' element type - T everywhere but in the multi-geom build's fp64 distance query, which runs on the fp32 tables: KS_MG_GJK_F64)
template <typename T, typename TV = T> struct PairGeo {
    T R1[9], p1[3], R2[9], p2[3];
    KS_TAB const TV* V1; KS_TAB const TV* V2;
    KS_TAB const unsigned short* off1; KS_TAB const unsigned short* adj1;
    KS_TAB const unsigned short* off2; KS_TAB const unsigned short* adj2;
    const unsigned short* dir1; const unsigned short* dir2;   // cube-map support start tables of the two hulls (global memory)
    int n1, n2;
    int hint1, hint2;          // last support vertex of each shape: start of the next hill climb
    T half_margin;
#ifdef KS_STAMP_HULL
    int cnt_support, cnt_steps;
    long long t_sup, t_clo;
#endif
};

// Hull vertex tables are stored padded: stride 4 reals (x, y, z, 0) and the count rounded up to a
// multiple of HULL_CHUNK with copies of vertex 0 (a copy never wins a strict arg-max / arg-min).
constexpr int HULL_CHUNK = 8;
#ifdef KS_SUPPORT_SKEW_OVERRIDE
constexpr double SUPPORT_SKEW = KS_SUPPORT_SKEW_OVERRIDE;
#else
constexpr double SUPPORT_SKEW = 1e-6;
#endif
constexpr double SKEW_X = 0.5377, SKEW_Y = -0.6240, SKEW_Z = 0.5671;   // see pair_support

// Support vertex of a convex hull along `dir` by hill climbing on the hull graph: from `hint`, move to
// the best strictly-improving neighbour until none improves.  On a convex polytope a vertex without an
// improving neighbour is a global maximiser, so this returns what the exhaustive scan of the oracle
// returns (they can differ only between exactly tied vertices).  Visits O(sqrt(V)) vertices instead of
// V (the palm hull has 754), and warm-started from the previous query usually only a handful.

// cube-map cell of a (hull-frame) direction: index into the hull's support start table
template <typename T> KS_HD int support_cell(const T* ld) {
    const T ax = kabs(ld[0]), ay = kabs(ld[1]), az = kabs(ld[2]);
    const int axis = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    const T major = axis == 0 ? ld[0] : (axis == 1 ? ld[1] : ld[2]);
    const T cu = axis == 0 ? ld[1] : (axis == 1 ? ld[2] : ld[0]), cv = axis == 0 ? ld[2] : (axis == 1 ? ld[0] : ld[1]);
    const T inv = T(0.5 * SUPPORT_R) / (kabs(major) > T(1e-30) ? kabs(major) : T(1e-30));
    int iu = (int)(cu * inv + T(0.5 * SUPPORT_R)), iv = (int)(cv * inv + T(0.5 * SUPPORT_R));
    iu = iu < 0 ? 0 : (iu > SUPPORT_R - 1 ? SUPPORT_R - 1 : iu);
    iv = iv < 0 ? 0 : (iv > SUPPORT_R - 1 ? SUPPORT_R - 1 : iv);
    return ((2 * axis + (major < 0 ? 1 : 0)) * SUPPORT_R + iu) * SUPPORT_R + iv;
}

// the climb: from the better of `hint` (the previous support vertex) and `tab` (the support vertex of the cube-map cell
// the direction falls in) to the support vertex along the hull-frame direction ld
template <typename T, typename TV>
KS_HD void hull_climb(const T* R, const T* p, KS_TAB const TV* V, KS_TAB const unsigned short* off, KS_TAB const unsigned short* adj, int tab, int& hint,
                      const T* ld, const T* dir, T hm, T* out) {
    int cur = hint;
    T best = V[4 * cur] * ld[0] + V[4 * cur + 1] * ld[1] + V[4 * cur + 2] * ld[2];
#ifndef KS_HINT_FIRST
    // start from the better of `hint` (the previous support vertex of this hull) and `tab` (the cell's support vertex, from global memory)
    bool tab_pending = false;
    {
        const T bt = V[4 * tab] * ld[0] + V[4 * tab + 1] * ld[1] + V[4 * tab + 2] * ld[2];
        if (bt > best) { best = bt; cur = tab; }
    }
#else
    // (Experiment, measured in round 4 and NOT kept: scan the neighbours of `hint` first and consult `tab` - several hundred cycles of L2
    // latency away - only when the climb has to move, once, after its first hop.  Same support vertex (a local maximiser is the global one);
    // A/B on one box: training 2.92 M against 3.04 M env-steps/s, sim-only 4.16 against 4.32 M, random-init protocol 3.38 against 3.49 M: the
    // directions of successive support queries differ enough that `hint` is rarely the answer, and the table read was already overlapped.)
    bool tab_pending = true;
#endif
    for (int guard = 0; guard < 4096; guard++) {
        const int c0 = off[cur], c1 = off[cur + 1];
        int nxt = cur;
        for (int c = c0; c < c1; c += 2) {
            // two chunks (eight neighbour ids) per round: both id reads are issued together, then the eight vertices, then
            // the comparisons in list order; a vertex with <= 4 neighbours left reads its last chunk twice (no effect)
            const int cb = c + 1 < c1 ? c + 1 : c;
            int j[8];
            T d[8];
            KS_UNROLL
            for (int q = 0; q < 4; q++) { j[q] = adj[4 * c + q]; j[4 + q] = adj[4 * cb + q]; }
            KS_UNROLL
            for (int q = 0; q < 8; q++) d[q] = V[4 * j[q]] * ld[0] + V[4 * j[q] + 1] * ld[1] + V[4 * j[q] + 2] * ld[2];
            KS_UNROLL
            for (int q = 0; q < 8; q++)
                if (d[q] > best) { best = d[q]; nxt = j[q]; }
        }
        if (nxt == cur) break;
        if (tab_pending) {
            tab_pending = false;
            const T bt = V[4 * tab] * ld[0] + V[4 * tab + 1] * ld[1] + V[4 * tab + 2] * ld[2];
            if (bt > best) { best = bt; nxt = tab; }
        }
        cur = nxt;
    }
    hint = cur;
    T v[3] = {V[4 * cur], V[4 * cur + 1], V[4 * cur + 2]};
    mulRv(out, R, v);
    add3(out, out, p);
    addscl3(out, dir, hm);
}

// exhaustive support of a small hull (n = padded count, a multiple of HULL_CHUNK; the padding repeats vertex 0 and never wins)
#ifndef KS_SCAN_MAX
#define KS_SCAN_MAX 0           // experiment (round 4), OFF: with 32 the cubes' 24 vertices are scanned - sim-only 4.24 M against 4.30 M with the climb
#endif
constexpr int SCAN_MAX = KS_SCAN_MAX;
template <typename T, typename TV>
KS_HD void hull_scan(const T* R, const T* p, KS_TAB const TV* V, int n, int& hint, const T* ld, const T* dir, T hm, T* out) {
    int cur = 0;
    T best = V[0] * ld[0] + V[1] * ld[1] + V[2] * ld[2];
    for (int i0 = 0; i0 < n; i0 += HULL_CHUNK) {
        T d[HULL_CHUNK];
        KS_UNROLL
        for (int q = 0; q < HULL_CHUNK; q++) d[q] = V[4 * (i0 + q)] * ld[0] + V[4 * (i0 + q) + 1] * ld[1] + V[4 * (i0 + q) + 2] * ld[2];
        KS_UNROLL
        for (int q = 0; q < HULL_CHUNK; q++)
            if (d[q] > best) { best = d[q]; cur = i0 + q; }
    }
    hint = cur;
    T v[3] = {V[4 * cur], V[4 * cur + 1], V[4 * cur + 2]};
    mulRv(out, R, v);
    add3(out, out, p);
    addscl3(out, dir, hm);
}

// Support points of BOTH hulls of a pair along dir / -dir.  The two cube-map reads (global memory, L2 resident, ~10x
// the latency of an LDS round) are issued first, back to back, so that the second one is in flight while the first
// hull is climbed.
template <typename T, typename TV> KS_HD void pair_support(PairGeo<T, TV>& g, const T* dir, T hm, T* out1, T* out2) {
    const T nd[3] = {-dir[0], -dir[1], -dir[2]};
    T ld1[3], ld2[3];
    mulRtv(ld1, g.R1, dir);
    mulRtv(ld2, g.R2, nd);
    // Tie rule shared with the oracle (ko_physics.c: hull_support): MPR and GJK ask for supports along the normals of faces
    // they built from the hulls' own vertices, so all of such a face's vertices attain the maximum to the last bit and rounding
    // would pick - and with it the portal path, the contact point on a flat feature, on a rounded polytope even the facet of the
    // normal.  The hull-frame direction is skewed by a fixed 1e-6 of its size: 1e-7 m of support error at most (a tenth of MPR's
    // tolerance), an order above the rounding of an fp32 direction, so that fp32 and fp64 mostly take the same vertex too.
    {
        const T s1 = T(SUPPORT_SKEW) * (kabs(ld1[0]) + kabs(ld1[1]) + kabs(ld1[2])), s2 = T(SUPPORT_SKEW) * (kabs(ld2[0]) + kabs(ld2[1]) + kabs(ld2[2]));
        ld1[0] += s1 * T(SKEW_X); ld1[1] += s1 * T(SKEW_Y); ld1[2] += s1 * T(SKEW_Z);
        ld2[0] += s2 * T(SKEW_X); ld2[1] += s2 * T(SKEW_Y); ld2[2] += s2 * T(SKEW_Z);
    }
    // (Experiment, off: SCAN_MAX = 0.)  A hull of at most SCAN_MAX vertices (the cubes: 24) scanned outright - its reads are independent
    // and pipeline through the LDS, a climb is a chain of dependent rounds plus a table read from L2; same vertex.  Measured slower: a warm
    // climb is one hop, and a wave whose lanes mix small and large hulls runs both code paths.
    const bool scan1 = g.n1 <= SCAN_MAX, scan2 = g.n2 <= SCAN_MAX;
    const int tab1 = scan1 ? 0 : g.dir1[support_cell(ld1)], tab2 = scan2 ? 0 : g.dir2[support_cell(ld2)];
    if (scan1) hull_scan(g.R1, g.p1, g.V1, g.n1, g.hint1, ld1, dir, hm, out1);
    else hull_climb(g.R1, g.p1, g.V1, g.off1, g.adj1, tab1, g.hint1, ld1, dir, hm, out1);
    if (scan2) hull_scan(g.R2, g.p2, g.V2, g.n2, g.hint2, ld2, nd, hm, out2);
    else hull_climb(g.R2, g.p2, g.V2, g.off2, g.adj2, tab2, g.hint2, ld2, nd, hm, out2);
}

// The Minkowski-difference point of a support pair.  EXPERIMENT of round 5 (KS_MINK_F64=1; default OFF): in fp32 v1 and v2 are world points of ~0.1 m, each
// rounded to 7.5e-9 - their difference carries 1e-8 of noise that differs from one support point of a query to the next, and MPR's portal normals (cross
// products of ~1 mm edge vectors between such points) turn it into ~5e-7 of noise on the dot products its 1e-6 tolerance tests decide on.  With the switch the
// difference is formed in fp64 from the two hull-frame vertices and the (fp32) poses - R1 a - R2 b + (p1 - p2) - and rounded once, at the magnitude of the
// difference itself.  Measured (host lane, tests/studies/divergence_table.py: 168 grasp-and-lift envs x 200 substeps within 1e-4 of the oracle): 104 -> 112;
// the WHOLE collision stage in fp64 on the same fp32 poses: 157.  On the GPU the switch costs k_rollout 49 more registers (256 + 153): the learner's 128-register
// waves would no longer fit beside it on a SIMD.  Not worth 8 envs: off.
#ifndef KS_MINK_F64
#define KS_MINK_F64 0
#endif
template <typename T, typename TV> KS_HD void minkowski_point_ids(const PairGeo<T, TV>& g, int i, int j, const T* v1, const T* v2, T* v);
template <typename T, typename TV> KS_HD void minkowski_point(const PairGeo<T, TV>& g, const T* v1, const T* v2, T* v) { minkowski_point_ids(g, g.hint1, g.hint2, v1, v2, v); }
template <typename T, typename TV> KS_HD void minkowski_point_ids(const PairGeo<T, TV>& g, int i, int j, const T* v1, const T* v2, T* v) {
    if constexpr (sizeof(T) == 4 && (KS_MINK_F64 != 0)) {
        const double a[3] = {(double)g.V1[4 * i], (double)g.V1[4 * i + 1], (double)g.V1[4 * i + 2]};
        const double b[3] = {(double)g.V2[4 * j], (double)g.V2[4 * j + 1], (double)g.V2[4 * j + 2]};
        KS_UNROLL
        for (int k = 0; k < 3; k++) {
            const double w1 = (double)g.R1[3 * k] * a[0] + (double)g.R1[3 * k + 1] * a[1] + (double)g.R1[3 * k + 2] * a[2];
            const double w2 = (double)g.R2[3 * k] * b[0] + (double)g.R2[3 * k + 1] * b[1] + (double)g.R2[3 * k + 2] * b[2];
            v[k] = (T)((w1 - w2) + ((double)g.p1[k] - (double)g.p2[k]));
        }
        (void)v1; (void)v2;
    } else {
        sub3(v, v1, v2);
    }
}

template <typename T> KS_HD void mpr_support(PairGeo<T>& g, const T* dir, Supp<T>& o) {
#ifdef KS_STAMP_HULL
    const int h1_ = g.hint1, h2_ = g.hint2;
#endif
    pair_support(g, dir, g.half_margin, o.v1, o.v2);
    o.i1 = g.hint1; o.i2 = g.hint2;
#ifdef KS_STAMP_HULL
    g.cnt_support += 2; g.cnt_steps += (h1_ != g.hint1) + (h2_ != g.hint2);
#endif
    if (g.half_margin == T(0)) minkowski_point(g, o.v1, o.v2, o.v);
    else sub3(o.v, o.v1, o.v2);
}

template <typename T> KS_HD bool is_zero(T x) { return kabs(x) < T(1e-15); }
template <typename T> KS_HD bool vec_is_zero(const T* v) { return is_zero(v[0]) && is_zero(v[1]) && is_zero(v[2]); }

template <typename T> KS_HD void portal_dir(const Supp<T>& v1, const Supp<T>& v2, const Supp<T>& v3, T* dir) {
    T a[3], b[3];
    sub3(a, v2.v, v1.v);
    sub3(b, v3.v, v1.v);
    cross3(dir, a, b);
    normalize3(dir);
}

template <typename T>
KS_HD bool portal_reach_tol(const Supp<T>& v1, const Supp<T>& v2, const Supp<T>& v3, const Supp<T>& v4, const T* dir, T tol) {
    T dv4 = dot3(v4.v, dir);
    T d1 = dv4 - dot3(v1.v, dir), d2 = dv4 - dot3(v2.v, dir), d3 = dv4 - dot3(v3.v, dir);
    T d = d1 < d2 ? d1 : d2;
    d = d < d3 ? d : d3;
    return is_zero(d) || d < tol;
}

template <typename T> KS_HD void assign(Supp<T>& d, const Supp<T>& s) {
    KS_UNROLL
    for (int i = 0; i < 3; i++) { d.v[i] = s.v[i]; d.v1[i] = s.v1[i]; d.v2[i] = s.v2[i]; }
    d.i1 = s.i1; d.i2 = s.i2;
}

template <typename T>
KS_HD void expand_portal(const Supp<T>& v0, Supp<T>& v1, Supp<T>& v2, Supp<T>& v3, const Supp<T>& v4) {
    T c[3];
    cross3(c, v4.v, v0.v);
    if (dot3(v1.v, c) > 0) {
        if (dot3(v2.v, c) > 0) assign(v1, v4); else assign(v3, v4);
    } else {
        if (dot3(v3.v, c) > 0) assign(v2, v4); else assign(v1, v4);
    }
}

template <typename T> KS_HD T origin_segment_dist2(const T* a, const T* b, T* wit) {
    T d[3];
    sub3(d, b, a);
    T t = -dot3(a, d), dd = dot3(d, d);
    if (t <= 0 || dd < T(1e-15)) copy3(wit, a);
    else if (t >= dd) copy3(wit, b);
    else { copy3(wit, a); addscl3(wit, d, t / dd); }
    return dot3(wit, wit);
}

template <typename T> KS_HD T origin_tri_dist2(const T* a, const T* b, const T* c, T* wit) {
    T d1[3], d2[3];
    sub3(d1, b, a);
    sub3(d2, c, a);
    T v = dot3(d1, d1), w = dot3(d2, d2), p = dot3(a, d1), q = dot3(a, d2), r = dot3(d1, d2);
    T den = w * v - r * r, sp = -1, tp = -1;
    if (!is_zero(den)) {
        sp = (q * r - w * p) / den;
        tp = (-sp * r - q) / w;
    }
    if (sp >= 0 && sp <= 1 && tp >= 0 && tp <= 1 && sp + tp <= 1) {
        copy3(wit, a);
        addscl3(wit, d1, sp);
        addscl3(wit, d2, tp);
        return dot3(wit, wit);
    }
    T w2[3], dist = origin_segment_dist2(a, b, wit), dd;
    dd = origin_segment_dist2(a, c, w2);
    if (dd < dist) { dist = dd; copy3(wit, w2); }
    dd = origin_segment_dist2(b, c, w2);
    if (dd < dist) { dist = dd; copy3(wit, w2); }
    return dist;
}

template <typename T>
KS_HD void find_pos(const Supp<T>& v0, const Supp<T>& v1, const Supp<T>& v2, const Supp<T>& v3, T* pos) {
    T dir[3], b[4], t[3], sum;
    portal_dir(v1, v2, v3, dir);
    cross3(t, v1.v, v2.v); b[0] = dot3(t, v3.v);
    cross3(t, v3.v, v2.v); b[1] = dot3(t, v0.v);
    cross3(t, v0.v, v1.v); b[2] = dot3(t, v3.v);
    cross3(t, v2.v, v1.v); b[3] = dot3(t, v0.v);
    sum = b[0] + b[1] + b[2] + b[3];
    if (is_zero(sum) || sum < 0) {
        b[0] = 0;
        cross3(t, v2.v, v3.v); b[1] = dot3(t, dir);
        cross3(t, v3.v, v1.v); b[2] = dot3(t, dir);
        cross3(t, v1.v, v2.v); b[3] = dot3(t, dir);
        sum = b[1] + b[2] + b[3];
    }
    T inv = T(1) / sum;
    KS_UNROLL
    for (int i = 0; i < 3; i++) {
        T p1 = b[0] * v0.v1[i] + b[1] * v1.v1[i] + b[2] * v2.v1[i] + b[3] * v3.v1[i];
        T p2 = b[0] * v0.v2[i] + b[1] * v1.v2[i] + b[2] * v2.v2[i] + b[3] * v3.v2[i];
        pos[i] = T(0.5) * (p1 + p2) * inv;
    }
}

// ---- What a lane remembers of a hull pair's last narrow-phase queries (lane-private: the pair -> lane dealing is fixed):
//   GJK: the vertex ids of the final simplex (<= 3 points).  The next query starts GJK from that simplex - the poses
//        have barely moved, so it is usually still the closest feature and the query ends after one confirming support
//        instead of ~10 iterations.
//   MPR: the vertex ids of the final portal.  While the origin ray still passes through it, the next query skips the
//        portal discovery and refines from there (a few supports instead of ~22).
// Four 32-bit words per pair: vertex ids are < 1024 (checked when the model is loaded), three ids per word, a count in the
// two top bits; word 0 = GJK ids on hull 1 + simplex size, 1 = GJK ids on hull 2, 2 = portal ids on hull 1 + 3 if valid,
// 3 = portal ids on hull 2.  All zeros = nothing remembered (cold start).  The stepping kernel keeps the words in registers
// across the substeps of a launch and carries them from one launch to the next through global memory (ks_api.hip), so the
// first substep of an env-step starts as warm as the other fourteen.
constexpr int WARM_WORDS = 4;
struct PairWarm {
    unsigned w[WARM_WORDS];
};
KS_HD unsigned pack3(int a, int b, int c, int top) { return (unsigned)a | ((unsigned)b << 10) | ((unsigned)c << 20) | ((unsigned)top << 30); }
// ids that do not fit the 10-bit fields (hulls of more than 1024 vertices, multi-geom build) are simply not remembered: a cold start
KS_HD bool packable(int a, int b, int c, int d, int e, int f) { return !MULTI_GEOM || (a | b | c | d | e | f) < 1024; }
KS_HD void unpack3(unsigned w, int* ids, int& top) {
    ids[0] = (int)(w & 1023u); ids[1] = (int)((w >> 10) & 1023u); ids[2] = (int)((w >> 20) & 1023u);
    top = (int)(w >> 30);
}

// Minkowski Portal Refinement penetration query (same decision structure as the oracle's mpr_penetration, i.e. the
// published algorithm of libccd's ccdMPRPenetration - libccd is (c) D. Fiser, BSD-3; it is a dependency of MuJoCo, not
// part of /root/reference, and no code of it is used here).  Returns true on overlap.
template <typename T, typename TV> KS_HD void hull_point(const T* R, const T* p, KS_TAB const TV* V, int i, T* out) {
    const T v[3] = {V[4 * i], V[4 * i + 1], V[4 * i + 2]};
    mulRv(out, R, v);
    add3(out, out, p);
}

// ---- Round 6: the penetration query of the fp32 PRODUCT (mpr_penetration_sm below; the template above stays the fp64 instantiation's - the
// parity instrument's - and the KS_MPR_SM=0 fall-back).
// (1) COLD again.  Rounds 3-5 started the product's query from the previous substep's final portal whenever that still was one (KS_MPR_WARM).
//     That is a different - equally valid - run of MPR, and on flat features it ends on another triangle of the same Minkowski facet: another
//     contact POINT.  The long-horizon parity tests never saw it: they step through ks_substep, whose queries are cold.  Measured through ks_step
//     in round 6 (tests/studies/long_horizon_envstep.py), the warm start left 76 of the 168 grasp-and-lift envs within 1e-4 of the oracle after
//     210 substeps; libccd's cold path - what MuJoCo runs, what the oracle follows and the recorded trajectory pins - leaves 141.
// (2) On fp64 Minkowski points.  Every point is formed from its two vertex IDS - fp32 hull tables and poses, fp64 arithmetic, as round 5's
//     read-off of the final portal - so portal normals, the supports along them, the stop test and the expansion's sign tests are decided as the
//     fp64 oracle decides them wherever both stand on the same portal: 162 of 168 through ks_step on the GPU and on the host lane
//     (tests/studies/divergence_table.py: variant r6).  State per portal corner: 3 doubles + 2 ids (SuppD).
// (3) ONE loop around ONE support site.  The template has five inlined support sites (v1, v2, discovery loop, "origin inside" loop, refinement
//     loop); the lanes of a wave that run a query are in different loops after the first two supports and the wave executes every site for as
//     many turns as ITS slowest lane needs there.  Here every lane, whatever phase its query is in, takes its next support in the same
//     instruction stream; the phases' own arithmetic (cross products, sign tests) is short and predicated.
// Measured and not kept (round 6): a PATH memory - the vertex pair each support of the previous substep's query returned, as the start of this
// substep's climbs (one confirming round per hull, no cube-map read): +1 % in training, -3 % sim-only, 8 MB of state; skipping the distance query
// for a pair that penetrated in the previous substep (KS_MPR_FIRST=3): nothing.  The cube map already puts a climb within a hop of its answer.
#ifndef KS_MPR_SM
#define KS_MPR_SM 1             // 0: the fp32 product runs the template above (fp32 points, cold unless KS_MPR_WARM=1): A/B and the divergence study
#endif
// The query is inlined into `collision` like the other narrow-phase queries.  That takes the stepping kernels' own register count past what leaves room
// for the learner's waves beside them (k_rollout 256 + 122 where 376 is the limit: training at 0.7 x) - the kernels' budgets are therefore SET in the
// source (ks_api.hip: KS_ROLLOUT_NUM_VGPR / KS_STEP_NUM_VGPR) and the excess is spilled.  Measured alternatives: the query out of line - every call
// saves 34 callee-saved registers of all 64 lanes for the two or three lanes that have a query, 110 MB of write-backs per env-step of 4096 envs
// (k_rollout's WRITE_SIZE 144 MB against 45 MB inlined), -1.7 % in training; the SUPPORT out of line - the pair record then lives in private
// memory -: sim-only 3.1 -> 2.5 M env-steps/s.
struct SuppD {
    double v[3];
    int i1, i2;
};
template <typename T> KS_HD void mink_f64(const PairGeo<T>& g, int i, int j, double* v) {
    const double a[3] = {(double)g.V1[4 * i], (double)g.V1[4 * i + 1], (double)g.V1[4 * i + 2]};
    const double b[3] = {(double)g.V2[4 * j], (double)g.V2[4 * j + 1], (double)g.V2[4 * j + 2]};
    KS_UNROLL
    for (int k = 0; k < 3; k++) {
        const double w1 = (double)g.R1[3 * k] * a[0] + (double)g.R1[3 * k + 1] * a[1] + (double)g.R1[3 * k + 2] * a[2];
        const double w2 = (double)g.R2[3 * k] * b[0] + (double)g.R2[3 * k + 1] * b[1] + (double)g.R2[3 * k + 2] * b[2];
        v[k] = (w1 - w2) + ((double)g.p1[k] - (double)g.p2[k]);
    }
}
// support vertex of a hull (fp32 table) along the hull-frame direction ld, fp64 dot products: hull_climb's rule (best strictly improving
// neighbour in list order) from `cur`
#ifndef KS_CLIMB64_WIDTH
#define KS_CLIMB64_WIDTH 8      // neighbours per round of the fp64 climb (hull_climb's two chunks at a time; 4: -2.6 % in training)
#endif
template <typename TV>
KS_HD int climb_f64(KS_TAB const TV* V, KS_TAB const unsigned short* off, KS_TAB const unsigned short* adj, int cur, int tab, const double* ld) {
    double best = (double)V[4 * cur] * ld[0] + (double)V[4 * cur + 1] * ld[1] + (double)V[4 * cur + 2] * ld[2];
    {
        const double bt = (double)V[4 * tab] * ld[0] + (double)V[4 * tab + 1] * ld[1] + (double)V[4 * tab + 2] * ld[2];
        if (bt > best) { best = bt; cur = tab; }
    }
    constexpr int W = KS_CLIMB64_WIDTH, CH = W / 4;
    for (int guard = 0; guard < 4096; guard++) {
        const int c0 = off[cur], c1 = off[cur + 1];
        int nxt = cur;
        for (int c = c0; c < c1; c += CH) {
            // W neighbour ids per round: id reads together, then the vertices, then the comparisons in list order (hull_climb's rule)
            int j[W];
            double d[W];
            KS_UNROLL
            for (int q = 0; q < W; q++) { const int cc = c + q / 4 < c1 ? c + q / 4 : c; j[q] = adj[4 * cc + (q & 3)]; }
            KS_UNROLL
            for (int q = 0; q < W; q++) d[q] = (double)V[4 * j[q]] * ld[0] + (double)V[4 * j[q] + 1] * ld[1] + (double)V[4 * j[q] + 2] * ld[2];
            KS_UNROLL
            for (int q = 0; q < W; q++)
                if (d[q] > best) { best = d[q]; nxt = j[q]; }
        }
        if (nxt == cur) break;
        cur = nxt;
    }
    return cur;
}
template <typename T> KS_HD void support_f64(PairGeo<T>& g, const double* d, SuppD& o) {
    double ld1[3], ld2[3];
    KS_UNROLL
    for (int k = 0; k < 3; k++) {
        ld1[k] = (double)g.R1[k] * d[0] + (double)g.R1[3 + k] * d[1] + (double)g.R1[6 + k] * d[2];
        ld2[k] = -((double)g.R2[k] * d[0] + (double)g.R2[3 + k] * d[1] + (double)g.R2[6 + k] * d[2]);
    }
    const double s1 = SUPPORT_SKEW * (kabs(ld1[0]) + kabs(ld1[1]) + kabs(ld1[2])), s2 = SUPPORT_SKEW * (kabs(ld2[0]) + kabs(ld2[1]) + kabs(ld2[2]));
    ld1[0] += s1 * SKEW_X; ld1[1] += s1 * SKEW_Y; ld1[2] += s1 * SKEW_Z;
    ld2[0] += s2 * SKEW_X; ld2[1] += s2 * SKEW_Y; ld2[2] += s2 * SKEW_Z;
    // the climbs start from the better of the last support vertex and the cube-map cell's (a start only: the climb decides in fp64)
    const T f1[3] = {(T)ld1[0], (T)ld1[1], (T)ld1[2]}, f2[3] = {(T)ld2[0], (T)ld2[1], (T)ld2[2]};
    const int tab1 = (int)g.dir1[support_cell(f1)], tab2 = (int)g.dir2[support_cell(f2)];
    g.hint1 = climb_f64(g.V1, g.off1, g.adj1, g.hint1, tab1, ld1);
    g.hint2 = climb_f64(g.V2, g.off2, g.adj2, g.hint2, tab2, ld2);
    o.i1 = g.hint1; o.i2 = g.hint2;
    mink_f64(g, o.i1, o.i2, o.v);
}
KS_HD void portal_dir_d(const SuppD& v1, const SuppD& v2, const SuppD& v3, double* dir) {
    double a[3], b[3];
    sub3(a, v2.v, v1.v);
    sub3(b, v3.v, v1.v);
    cross3(dir, a, b);
    normalize3(dir);
}
KS_HD bool portal_reach_tol_d(const SuppD& v1, const SuppD& v2, const SuppD& v3, const SuppD& v4, const double* dir, double tol) {
    const double dv4 = dot3(v4.v, dir);
    const double d1 = dv4 - dot3(v1.v, dir), d2 = dv4 - dot3(v2.v, dir), d3 = dv4 - dot3(v3.v, dir);
    double d = d1 < d2 ? d1 : d2;
    d = d < d3 ? d : d3;
    return is_zero(d) || d < tol;
}
KS_HD void expand_portal_d(const SuppD& v0, SuppD& v1, SuppD& v2, SuppD& v3, const SuppD& v4) {
    double c[3];
    cross3(c, v4.v, v0.v);
    if (dot3(v1.v, c) > 0) {
        if (dot3(v2.v, c) > 0) v1 = v4; else v3 = v4;
    } else {
        if (dot3(v3.v, c) > 0) v2 = v4; else v1 = v4;
    }
}
// depth / direction / contact point read off the final portal (fp64 points; the hulls' own support points rebuilt from the vertex ids)
template <typename T>
KS_HD bool mpr_readoff_f64(const PairGeo<T>& g, const SuppD& v0, const SuppD& v1, const SuppD& v2, const SuppD& v3, T* depth, T* dir, T* pos) {
    double wit[3], d[3];
    const double dd = origin_tri_dist2<double>(v1.v, v2.v, v3.v, wit);
    *depth = (T)std::sqrt(dd);
    if (vec_is_zero(wit)) return false;
    const double nn = std::sqrt(wit[0] * wit[0] + wit[1] * wit[1] + wit[2] * wit[2]);
    dir[0] = (T)(wit[0] / nn); dir[1] = (T)(wit[1] / nn); dir[2] = (T)(wit[2] / nn);
    double b[4], t[3], sum;
    portal_dir_d(v1, v2, v3, d);
    cross3(t, v1.v, v2.v); b[0] = dot3(t, v3.v);
    cross3(t, v3.v, v2.v); b[1] = dot3(t, v0.v);
    cross3(t, v0.v, v1.v); b[2] = dot3(t, v3.v);
    cross3(t, v2.v, v1.v); b[3] = dot3(t, v0.v);
    sum = b[0] + b[1] + b[2] + b[3];
    if (is_zero(sum) || sum < 0) {
        b[0] = 0;
        cross3(t, v2.v, v3.v); b[1] = dot3(t, d);
        cross3(t, v3.v, v1.v); b[2] = dot3(t, d);
        cross3(t, v1.v, v2.v); b[3] = dot3(t, d);
        sum = b[1] + b[2] + b[3];
    }
    const double inv = 1.0 / sum;
    const SuppD* sv[3] = {&v1, &v2, &v3};
    double acc[3] = {b[0] * ((double)g.p1[0] + (double)g.p2[0]), b[0] * ((double)g.p1[1] + (double)g.p2[1]), b[0] * ((double)g.p1[2] + (double)g.p2[2])};
    for (int q = 0; q < 3; q++) {
        const int i = sv[q]->i1, j = sv[q]->i2;
        KS_UNROLL
        for (int k = 0; k < 3; k++) {
            const double w1 = (double)g.R1[3 * k] * (double)g.V1[4 * i] + (double)g.R1[3 * k + 1] * (double)g.V1[4 * i + 1] + (double)g.R1[3 * k + 2] * (double)g.V1[4 * i + 2] + (double)g.p1[k];
            const double w2 = (double)g.R2[3 * k] * (double)g.V2[4 * j] + (double)g.R2[3 * k + 1] * (double)g.V2[4 * j + 1] + (double)g.R2[3 * k + 2] * (double)g.V2[4 * j + 2] + (double)g.p2[k];
            acc[k] += b[q + 1] * (w1 + w2);
        }
    }
    KS_UNROLL
    for (int k = 0; k < 3; k++) pos[k] = (T)(0.5 * acc[k] * inv);
    return true;
}

// ---- The penetration query of the fp32 product (round 6): libccd's cold path, decision for decision, as ONE loop around ONE support call.
// mpr_penetration above has five inlined support sites (v1, v2, the discovery loop, the "origin inside" loop, the refinement loop); the lanes of a
// wave that run a query are in different loops after the first two supports, and the wave executes every site for as many turns as ITS slowest
// lane needs there.  A support pair - two hill climbs, chains of dependent LDS reads - is ~6 k cycles of a wave that has nothing else to issue, a
// cold query is ~10 of them.  Here every lane, whatever phase its query is in, takes its next support in the same instruction stream: the wave runs
// max(supports of a lane) turns instead of the sum over the sites of the per-site maxima.  The phases' own arithmetic (cross products, sign tests)
// is short and predicated.  All of it on fp64 Minkowski points formed from vertex ids (SuppD): same decisions as the fp64 oracle wherever both
// stand on the same portal (tests/studies/divergence_table.py: 162 of 168 grasp-and-lift envs within 1e-4 after 200 substeps; fp32 points: 146).
enum { MPR_S_V1 = 0, MPR_S_V2 = 1, MPR_S_V3 = 2, MPR_S_INSIDE = 3, MPR_S_REFINE = 4 };
template <typename T>
KS_NARROW bool mpr_penetration_sm(PairGeo<T>& g_io, T tol_, int max_iter, T* depth_o, T* dir_o, T* pos_o) {
    // (the pair record and the result slots are worked on in local copies and written once at the end)
    PairGeo<T> g = g_io;
    T depth_[1] = {0}, dir[3] = {0, 0, 0}, pos[3] = {0, 0, 0};
    T* depth = depth_;
    struct Out {
        PairGeo<T>& g_io; const PairGeo<T>& g; T* depth_o; T* dir_o; T* pos_o; const T* depth; const T* dir; const T* pos;
        KS_HD ~Out() {
#ifdef KS_STAMP_HULL
            g_io.t_clo = g.t_clo; g_io.cnt_support = g.cnt_support;
#endif
            g_io.hint1 = g.hint1; g_io.hint2 = g.hint2; *depth_o = *depth; dir_o[0] = dir[0]; dir_o[1] = dir[1]; dir_o[2] = dir[2]; pos_o[0] = pos[0]; pos_o[1] = pos[1]; pos_o[2] = pos[2];
        }
    } out_{g_io, g, depth_o, dir_o, pos_o, depth, dir, pos};
    SuppD v0, v1, v2, v3, v4;
    double d[3], va[3], vb[3];
    const double tol = (double)tol_;
    KS_UNROLL
    for (int k = 0; k < 3; k++) v0.v[k] = (double)g.p1[k] - (double)g.p2[k];
    if (vec_is_zero(v0.v)) v0.v[0] += 1e-5;
    v0.i1 = 0; v0.i2 = 0;
    v1 = v0; v2 = v0; v3 = v0;
    KS_UNROLL
    for (int k = 0; k < 3; k++) d[k] = -v0.v[k];
    normalize3(d);
    int state = MPR_S_V1, it = 0;
    for (;;) {
        if (state >= MPR_S_INSIDE) {
            // the portal's normal; "origin inside the portal" ends when the portal faces the origin - the refinement then starts on the same normal
            portal_dir_d(v1, v2, v3, d);
            if (state == MPR_S_INSIDE) {
                if (it > 100) return false;
                const double dt = dot3(d, v1.v);
                if (is_zero(dt) || dt > 0) { state = MPR_S_REFINE; it = 0; }
            }
        } else if (state == MPR_S_V3 && it > 100) return false;
#if defined(KS_STAMP_HULL) && defined(__HIP_DEVICE_COMPILE__)
        const long long tsup0_ = clock64();
#endif
        support_f64(g, d, v4);                                  // THE support site
#if defined(KS_STAMP_HULL) && defined(__HIP_DEVICE_COMPILE__)
        g.t_clo += clock64() - tsup0_; g.cnt_support += 2;        // (diagnostic: cycles inside the supports of this query)
#endif
        const double dv = dot3(v4.v, d);
        if (state == MPR_S_V1) {
            v1 = v4;
            if (is_zero(dv) || dv < 0) return false;
            cross3(d, v0.v, v1.v);
            if (vec_is_zero(d)) {
                if (vec_is_zero(v1.v)) return false;
                const double nn = std::sqrt(dot3(v1.v, v1.v));
                *depth = (T)nn;
                KS_UNROLL
                for (int k = 0; k < 3; k++) dir[k] = (T)(v1.v[k] / nn);
                T a[3], b[3];
                hull_point(g.R1, g.p1, g.V1, v1.i1, a); hull_point(g.R2, g.p2, g.V2, v1.i2, b);
                KS_UNROLL
                for (int k = 0; k < 3; k++) pos[k] = T(0.5) * (a[k] + b[k]);
                return true;
            }
            normalize3(d);
            state = MPR_S_V2;
        } else if (state == MPR_S_V2) {
            v2 = v4;
            if (is_zero(dv) || dv < 0) return false;
            sub3(va, v1.v, v0.v);
            sub3(vb, v2.v, v0.v);
            cross3(d, va, vb);
            normalize3(d);
            if (dot3(d, v0.v) > 0) {
                const SuppD t = v1; v1 = v2; v2 = t;
                d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2];
            }
            state = MPR_S_V3; it = 0;
        } else if (state == MPR_S_V3) {
            v3 = v4;
            if (is_zero(dv) || dv < 0) return false;
            bool cont = false;
            cross3(va, v1.v, v3.v);
            double dt = dot3(va, v0.v);
            if (dt < 0 && !is_zero(dt)) { v2 = v3; cont = true; }
            if (!cont) {
                cross3(va, v3.v, v2.v);
                dt = dot3(va, v0.v);
                if (dt < 0 && !is_zero(dt)) { v1 = v3; cont = true; }
            }
            if (cont) {
                sub3(va, v1.v, v0.v);
                sub3(vb, v2.v, v0.v);
                cross3(d, va, vb);
                normalize3(d);
                it++;
            } else { state = MPR_S_INSIDE; it = 0; }
        } else if (state == MPR_S_INSIDE) {
            if (!(is_zero(dv) || dv > 0)) return false;
            if (portal_reach_tol_d(v1, v2, v3, v4, d, tol)) return false;
            expand_portal_d(v0, v1, v2, v3, v4);
            it++;
        } else {
            if (portal_reach_tol_d(v1, v2, v3, v4, d, tol) || it > max_iter) return mpr_readoff_f64(g, v0, v1, v2, v3, depth, dir, pos);
            expand_portal_d(v0, v1, v2, v3, v4);
            it++;
        }
    }
}

// ---- The same query on TWO lanes (round 6, fp32 product on the GPU, 16-lane teams): the lane that owns the pair and the team lane 8 places
// away (sub ^ 8, one DPP row_ror:8 mov per word).  A turn of the query above is ~8 k cycles of a wave in which two or three lanes work: the instruction
// stream is issue-bound (one wave per SIMD: ~5 cycles per vector instruction, 64 per dependent LDS read - tools/r06/ubench/lat.hip), and more than
// half of a turn is its support - two hill climbs, one per hull, one after the other.  Here the OWNER climbs hull 1 and the HELPER hull 2 in the same
// instructions: each lane holds ONE hull (pose, tables, last support vertex), forms its own hull-frame direction, reads its own cube-map cell, climbs,
// and forms its own world point R V[i] in fp64; the two exchange point and vertex id, and both form the Minkowski point (w1 - w2) + (p1 - p2) - the
// helper as -(w2 - w1), the same bits.  The state machine around the support runs on BOTH lanes, on the same values: every decision falls alike, the
// pair leaves the loop together, and the owner reads the result off the final portal (the helper's hull re-fetched by DPP).  Bit-identical to
// mpr_penetration_sm by construction: the same operations on the same operands in the same order, on another lane.
// Both lanes of a pair must be active in every exchange: the caller brings them here together (collide_hull_hull_split).
#if defined(__HIP_DEVICE_COMPILE__)
template <typename X> __device__ __forceinline__ X dpp_partner(X x) {      // x of team lane sub ^ 8
    if constexpr (sizeof(X) == 4) return __builtin_bit_cast(X, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));
    else {
        static_assert(sizeof(X) == 8, "32- or 64-bit values");
        struct P2 { int lo, hi; };
        P2 q = __builtin_bit_cast(P2, x);
        q.lo = __builtin_amdgcn_update_dpp(0, q.lo, 0x128, 0xf, 0xf, false);
        q.hi = __builtin_amdgcn_update_dpp(0, q.hi, 0x128, 0xf, 0xf, false);
        return __builtin_bit_cast(X, q);
    }
}
// One hull of a pair as the lane that climbs it holds it
template <typename T> struct HullGeo {
    T R[9], p[3], po[3];                // pose of MY hull, position of the OTHER one
    KS_TAB const T* V; KS_TAB const unsigned short* off; KS_TAB const unsigned short* adj;
    const unsigned short* dir;
    int hint;
};
// The portal's three corners live in LDS (24 words per lane in the env's contact region, dead until the collision stage merges its
// staged records).  In mpr_penetration_sm the compiler keeps v1 .. v3 in PRIVATE memory (expand_portal_d's `v1 = v4 / v3 = v4` becomes one store
// through a selected pointer): every turn stores a corner and the next turn loads all three back - a round trip through the vector memory path
// (a written line is not held by the L1) where an LDS read costs 64 cycles.  Forcing them into registers was measured slower (the state machine's
// merges then copy 24 registers per path: 2.37 -> 2.24 M env-steps/s); the two-lane query with its portal in private memory: the same bits,
// sim-only 3.50 against 3.55 M.  Same operations on the same operands: bit-identical.
struct PortalLds {
    KS_LDS double* b;       // corner c (0: v1, 1: v2, 2: v3) at b + 4 c: v[3], then the two vertex ids in the fourth slot
    __device__ __forceinline__ void getv(int c, double* v) const { v[0] = b[4 * c]; v[1] = b[4 * c + 1]; v[2] = b[4 * c + 2]; }
    __device__ __forceinline__ SuppD get(int c) const {
        SuppD s;
        getv(c, s.v);
        KS_LDS const int* q = (KS_LDS const int*)(b + 4 * c + 3);
        s.i1 = q[0]; s.i2 = q[1];
        return s;
    }
    __device__ __forceinline__ void put(int c, const SuppD& s) const {
        b[4 * c] = s.v[0]; b[4 * c + 1] = s.v[1]; b[4 * c + 2] = s.v[2];
        KS_LDS int* q = (KS_LDS int*)(b + 4 * c + 3);
        q[0] = s.i1; q[1] = s.i2;
    }
};
template <typename T>
__device__ __forceinline__ bool mpr_penetration_pair(const PairGeo<T>& g_own, bool own, KS_LDS double* portal, T tol_, int max_iter, int& hint1_o, int& hint2_o, T* depth_o,
                                                         T* dir_o, T* pos_o, int* turns_o = nullptr) {
    HullGeo<T> h;
    KS_UNROLL
    for (int k = 0; k < 9; k++) { const T o = dpp_partner(g_own.R2[k]); h.R[k] = own ? g_own.R1[k] : o; }
    KS_UNROLL
    for (int k = 0; k < 3; k++) {
        const T o2 = dpp_partner(g_own.p2[k]), o1 = dpp_partner(g_own.p1[k]);
        h.p[k] = own ? g_own.p1[k] : o2;
        h.po[k] = own ? g_own.p2[k] : o1;
    }
    { const auto o = dpp_partner(g_own.V2); h.V = own ? g_own.V1 : o; }
    { const auto o = dpp_partner(g_own.off2); h.off = own ? g_own.off1 : o; }
    { const auto o = dpp_partner(g_own.adj2); h.adj = own ? g_own.adj1 : o; }
    { const auto o = dpp_partner(g_own.dir2); h.dir = own ? g_own.dir1 : o; }
    { const int o = dpp_partner(g_own.hint2); h.hint = own ? g_own.hint1 : o; }
    const PortalLds P{portal};
    const double sgn = own ? 1.0 : -1.0;
    SuppD v0, v4;
    double d[3], va[3], vb[3], c1[3], c2[3], c3[3];
    const double tol = (double)tol_;
    KS_UNROLL
    for (int k = 0; k < 3; k++) v0.v[k] = sgn * ((double)h.p[k] - (double)h.po[k]);
    if (vec_is_zero(v0.v)) v0.v[0] += 1e-5;
    v0.i1 = 0; v0.i2 = 0;
    P.put(0, v0); P.put(1, v0); P.put(2, v0);
    KS_UNROLL
    for (int k = 0; k < 3; k++) d[k] = -v0.v[k];
    normalize3(d);
    int state = MPR_S_V1, it = 0;
    bool hit = false;
    for (;;) {
#ifdef KS_STAMP_SPLIT
        if (turns_o) (*turns_o)++;
#endif
        if (state >= MPR_S_INSIDE) {
            P.getv(0, c1); P.getv(1, c2); P.getv(2, c3);
            sub3(va, c2, c1);
            sub3(vb, c3, c1);
            cross3(d, va, vb);
            normalize3(d);
            if (state == MPR_S_INSIDE) {
                if (it > 100) break;
                const double dt = dot3(d, c1);
                if (is_zero(dt) || dt > 0) { state = MPR_S_REFINE; it = 0; }
            }
        } else if (state == MPR_S_V3 && it > 100) break;
        {
            double ld[3];
            KS_UNROLL
            for (int k = 0; k < 3; k++) {
                const double a = (double)h.R[k] * d[0] + (double)h.R[3 + k] * d[1] + (double)h.R[6 + k] * d[2];
                ld[k] = own ? a : -a;
            }
            const double s1 = SUPPORT_SKEW * (kabs(ld[0]) + kabs(ld[1]) + kabs(ld[2]));
            ld[0] += s1 * SKEW_X; ld[1] += s1 * SKEW_Y; ld[2] += s1 * SKEW_Z;
            const T f1[3] = {(T)ld[0], (T)ld[1], (T)ld[2]};
            const int tab = (int)h.dir[support_cell(f1)];
            h.hint = climb_f64(h.V, h.off, h.adj, h.hint, tab, ld);
            const double a[3] = {(double)h.V[4 * h.hint], (double)h.V[4 * h.hint + 1], (double)h.V[4 * h.hint + 2]};
            const int io = dpp_partner(h.hint);
            v4.i1 = own ? h.hint : io; v4.i2 = own ? io : h.hint;
            KS_UNROLL
            for (int k = 0; k < 3; k++) {
                const double w = (double)h.R[3 * k] * a[0] + (double)h.R[3 * k + 1] * a[1] + (double)h.R[3 * k + 2] * a[2];
                const double wo = dpp_partner(w);
                v4.v[k] = sgn * (w - wo) + sgn * ((double)h.p[k] - (double)h.po[k]);
            }
        }
        const double dv = dot3(v4.v, d);
        if (state == MPR_S_V1) {
            P.put(0, v4);                                           // v1 = v4
            if (is_zero(dv) || dv < 0) break;
            cross3(d, v0.v, v4.v);
            if (vec_is_zero(d)) {
                if (vec_is_zero(v4.v)) break;
                const double nn = std::sqrt(dot3(v4.v, v4.v));
                *depth_o = (T)nn;
                KS_UNROLL
                for (int k = 0; k < 3; k++) dir_o[k] = (T)(v4.v[k] / nn);
                T a[3], b[3];
                hull_point(h.R, h.p, h.V, h.hint, a);
                KS_UNROLL
                for (int k = 0; k < 3; k++) b[k] = dpp_partner(a[k]);
                KS_UNROLL
                for (int k = 0; k < 3; k++) pos_o[k] = T(0.5) * (a[k] + b[k]);
                hit = true;
                break;
            }
            normalize3(d);
            state = MPR_S_V2;
        } else if (state == MPR_S_V2) {
            if (is_zero(dv) || dv < 0) { P.put(1, v4); break; }
            const SuppD o1 = P.get(0);
            sub3(va, o1.v, v0.v);
            sub3(vb, v4.v, v0.v);
            cross3(d, va, vb);
            normalize3(d);
            if (dot3(d, v0.v) > 0) {
                P.put(0, v4); P.put(1, o1);                         // v1 <-> v2
                d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2];
            } else P.put(1, v4);                                    // v2 = v4
            state = MPR_S_V3; it = 0;
        } else if (state == MPR_S_V3) {
            if (is_zero(dv) || dv < 0) break;
            bool cont = false;
            P.getv(0, c1); P.getv(1, c2);
            cross3(va, c1, v4.v);
            double dt = dot3(va, v0.v);
            if (dt < 0 && !is_zero(dt)) { P.put(1, v4); copy3(c2, v4.v); cont = true; }                 // v2 = v3
            if (!cont) {
                cross3(va, v4.v, c2);
                dt = dot3(va, v0.v);
                if (dt < 0 && !is_zero(dt)) { P.put(0, v4); copy3(c1, v4.v); cont = true; }             // v1 = v3
            }
            if (cont) {
                sub3(va, c1, v0.v);
                sub3(vb, c2, v0.v);
                cross3(d, va, vb);
                normalize3(d);
                it++;
            } else { P.put(2, v4); state = MPR_S_INSIDE; it = 0; }                                      // v3 = v4
        } else {
            if (state == MPR_S_INSIDE && !(is_zero(dv) || dv > 0)) break;
            P.getv(0, c1); P.getv(1, c2); P.getv(2, c3);
            // portal_reach_tol_d
            bool reached;
            {
                const double dv4 = dot3(v4.v, d);
                const double d1 = dv4 - dot3(c1, d), d2 = dv4 - dot3(c2, d), d3 = dv4 - dot3(c3, d);
                double dm = d1 < d2 ? d1 : d2;
                dm = dm < d3 ? dm : d3;
                reached = is_zero(dm) || dm < tol;
            }
            if (state == MPR_S_INSIDE) { if (reached) break; }
            else if (reached || it > max_iter) {
                PairGeo<T> g;
                KS_UNROLL
                for (int k = 0; k < 9; k++) { const T o = dpp_partner(h.R[k]); g.R1[k] = own ? h.R[k] : o; g.R2[k] = own ? o : h.R[k]; }
                KS_UNROLL
                for (int k = 0; k < 3; k++) { g.p1[k] = own ? h.p[k] : h.po[k]; g.p2[k] = own ? h.po[k] : h.p[k]; }
                { const auto o = dpp_partner(h.V); g.V1 = own ? h.V : o; g.V2 = own ? o : h.V; }
                const SuppD q1 = P.get(0), q2 = P.get(1), q3 = P.get(2);
                hit = mpr_readoff_f64(g, v0, q1, q2, q3, depth_o, dir_o, pos_o);
                break;
            }
            // expand_portal_d
            {
                double cc[3];
                cross3(cc, v4.v, v0.v);
                const int w = dot3(c1, cc) > 0 ? (dot3(c2, cc) > 0 ? 0 : 2) : (dot3(c3, cc) > 0 ? 1 : 0);
                P.put(w, v4);
            }
            it++;
        }
    }
    { const int o = dpp_partner(h.hint); hint1_o = own ? h.hint : o; hint2_o = own ? o : h.hint; }
    return hit;
}
#endif

template <typename T>
KS_NARROW bool mpr_penetration(PairGeo<T>& g, T tol, int max_iter, T* depth, T* dir, T* pos, PairWarm* ws = nullptr) {
    Supp<T> v0, v1, v2, v3, v4;
    T d[3], va[3], vb[3];
    copy3(v0.v1, g.p1);
    copy3(v0.v2, g.p2);
    sub3(v0.v, v0.v1, v0.v2);
    if (vec_is_zero(v0.v)) v0.v[0] += T(1e-5);
    v0.i1 = 0; v0.i2 = 0;
    bool have_portal = false;
    int ma[3] = {0, 0, 0}, mb[3] = {0, 0, 0}, mn = 0, unused = 0;
    // The portal MPR ends on - hence depth, normal and above all the contact POINT on flat features - depends on the path within
    // its 1e-6 tolerance.  The cold path below is libccd's, which the oracle follows and which reproduces real MuJoCo 1.50 to 1e-9
    // through 18 rows of contact (tests/test_mujoco_recorded.py): the fp64 instantiation (the parity instrument) always takes it.
    // The fp32 product starts from the previous substep's portal when that still is one (KS_MPR_WARM, default on): fp32 rounding
    // of the supports' near-ties changes the path anyway, and a warm query is 1 - 2 support pairs instead of 6 - 10.
#ifndef KS_MPR_WARM
#define KS_MPR_WARM 0           // (round 6: off - see the path memory above; 1 restores rounds 3-5's warm start for A/B)
#endif
    constexpr bool mpr_warm = (KS_MPR_WARM != 0) && sizeof(T) == 4;
    if (mpr_warm && ws != nullptr) { unpack3(ws->w[2], ma, mn); unpack3(ws->w[3], mb, unused); }
    if (mn == 3) {
        // the previous query's portal at the current poses: still a portal if the origin ray (from v0 through the
        // origin) passes through the triangle, i.e. the origin is on the inner side of the three planes (v0, vi, vj)
        v1.i1 = ma[0]; v1.i2 = mb[0]; v2.i1 = ma[1]; v2.i2 = mb[1]; v3.i1 = ma[2]; v3.i2 = mb[2];
        hull_point(g.R1, g.p1, g.V1, ma[0], v1.v1); hull_point(g.R2, g.p2, g.V2, mb[0], v1.v2); minkowski_point_ids(g, ma[0], mb[0], v1.v1, v1.v2, v1.v);
        hull_point(g.R1, g.p1, g.V1, ma[1], v2.v1); hull_point(g.R2, g.p2, g.V2, mb[1], v2.v2); minkowski_point_ids(g, ma[1], mb[1], v2.v1, v2.v2, v2.v);
        hull_point(g.R1, g.p1, g.V1, ma[2], v3.v1); hull_point(g.R2, g.p2, g.V2, mb[2], v3.v2); minkowski_point_ids(g, ma[2], mb[2], v3.v1, v3.v2, v3.v);
        T c13[3], c32[3], c21[3], e1[3], e2[3], nn[3];
        cross3(c13, v1.v, v3.v); cross3(c32, v3.v, v2.v); cross3(c21, v2.v, v1.v);
        sub3(e1, v2.v, v1.v); sub3(e2, v3.v, v1.v); cross3(nn, e1, e2);
        have_portal = dot3(c13, v0.v) >= 0 && dot3(c32, v0.v) >= 0 && dot3(c21, v0.v) >= 0 && dot3(nn, nn) > T(1e-24);
        ws->w[2] = 0;                                   // valid again only if this query ends on a portal
    }
    if (!have_portal) {
    scl3(d, v0.v, T(-1));
    normalize3(d);
    mpr_support(g, d, v1);
    T dt = dot3(v1.v, d);
    if (is_zero(dt) || dt < 0) return false;
    cross3(d, v0.v, v1.v);
    if (vec_is_zero(d)) {
        if (vec_is_zero(v1.v)) return false;
        *depth = norm3(v1.v);
        copy3(dir, v1.v);
        normalize3(dir);
        KS_UNROLL
        for (int i = 0; i < 3; i++) pos[i] = T(0.5) * (v1.v1[i] + v1.v2[i]);
        return true;
    }
    normalize3(d);
    mpr_support(g, d, v2);
    dt = dot3(v2.v, d);
    if (is_zero(dt) || dt < 0) return false;
    sub3(va, v1.v, v0.v);
    sub3(vb, v2.v, v0.v);
    cross3(d, va, vb);
    normalize3(d);
    if (dot3(d, v0.v) > 0) {
        Supp<T> t;
        assign(t, v1); assign(v1, v2); assign(v2, t);
        scl3(d, d, T(-1));
    }
    for (int it = 0;; it++) {
        if (it > 100) return false;
        mpr_support(g, d, v3);
        dt = dot3(v3.v, d);
        if (is_zero(dt) || dt < 0) return false;
        bool cont = false;
        cross3(va, v1.v, v3.v);
        dt = dot3(va, v0.v);
        if (dt < 0 && !is_zero(dt)) { assign(v2, v3); cont = true; }
        if (!cont) {
            cross3(va, v3.v, v2.v);
            dt = dot3(va, v0.v);
            if (dt < 0 && !is_zero(dt)) { assign(v1, v3); cont = true; }
        }
        if (!cont) break;
        sub3(va, v1.v, v0.v);
        sub3(vb, v2.v, v0.v);
        cross3(d, va, vb);
        normalize3(d);
    }
    }   // portal discovery
    T dt;
    for (int it = 0;; it++) {
        if (it > 100) return false;
        portal_dir(v1, v2, v3, d);
        dt = dot3(d, v1.v);
        if (is_zero(dt) || dt > 0) break;
        mpr_support(g, d, v4);
        dt = dot3(v4.v, d);
        if (!(is_zero(dt) || dt > 0)) return false;
        if (portal_reach_tol(v1, v2, v3, v4, d, tol)) return false;
        expand_portal(v0, v1, v2, v3, v4);
    }
    for (int it = 0;; it++) {
        portal_dir(v1, v2, v3, d);
        mpr_support(g, d, v4);
        if (portal_reach_tol(v1, v2, v3, v4, d, tol) || it > max_iter) {
            T wit[3];
#ifndef KS_REFINE_F64
#define KS_REFINE_F64 1        // round 5, fp32 product: depth and direction of the FINAL portal recomputed in fp64 from its vertex ids (0: round 4's fp32 read-off)
#endif
            if constexpr (sizeof(T) == 4 && (KS_REFINE_F64 != 0)) {
                // The portal is what the fp32 iteration ended on; what is READ OFF it - the distance of the origin from a triangle of ~1 mm edges
                // ~0.03 m out in the Minkowski difference - lost ~1e-7 m to the fp32 rounding of the triangle's normal: a depth error of 0.1 % that
                // does not average out (the same vertices for many substeps: a bias, x 2770 / s^2 of contact stiffness) and that decided more of the
                // fp32 product's long-horizon drift from the oracle than every discrete event together (tests/studies/divergence_table.py: 104 -> 129
                // of 168 grasp-and-lift envs within 1e-4 after 200 substeps on the host lane; an all-fp64 collision stage: 157).  Same three vertex
                // pairs, fp64 arithmetic on the fp32 tables and poses, once per penetrating pair and substep; no register is added to k_rollout.
                double P[3][3];
                const Supp<T>* sv[3] = {&v1, &v2, &v3};
                for (int q = 0; q < 3; q++) {
                    const int i = sv[q]->i1, j = sv[q]->i2;
                    for (int k = 0; k < 3; k++) {
                        const double w1 = (double)g.R1[3 * k] * (double)g.V1[4 * i] + (double)g.R1[3 * k + 1] * (double)g.V1[4 * i + 1] + (double)g.R1[3 * k + 2] * (double)g.V1[4 * i + 2];
                        const double w2 = (double)g.R2[3 * k] * (double)g.V2[4 * j] + (double)g.R2[3 * k + 1] * (double)g.V2[4 * j + 1] + (double)g.R2[3 * k + 2] * (double)g.V2[4 * j + 2];
                        P[q][k] = (w1 - w2) + ((double)g.p1[k] - (double)g.p2[k]);
                    }
                }
                double witd[3];
                const double dd = origin_tri_dist2<double>(P[0], P[1], P[2], witd);
                *depth = (T)std::sqrt(dd);
                wit[0] = (T)witd[0]; wit[1] = (T)witd[1]; wit[2] = (T)witd[2];
                if (vec_is_zero(wit)) return false;
                const double nn = std::sqrt(witd[0] * witd[0] + witd[1] * witd[1] + witd[2] * witd[2]);
                dir[0] = (T)(witd[0] / nn); dir[1] = (T)(witd[1] / nn); dir[2] = (T)(witd[2] / nn);
                find_pos(v0, v1, v2, v3, pos);
                goto refined_;
            }
            *depth = ksqrt(origin_tri_dist2(v1.v, v2.v, v3.v, wit));
            if (vec_is_zero(wit)) return false;
            copy3(dir, wit);
            normalize3(dir);
            find_pos(v0, v1, v2, v3, pos);
        refined_:
            if (mpr_warm && ws != nullptr && packable(v1.i1, v2.i1, v3.i1, v1.i2, v2.i2, v3.i2)) {
                ws->w[2] = pack3(v1.i1, v2.i1, v3.i1, 3);
                ws->w[3] = pack3(v1.i2, v2.i2, v3.i2, 0);
            }
            return true;
        }
        expand_portal(v0, v1, v2, v3, v4);
    }
}


// ---- GJK closest-features query on the un-inflated hulls (margin-zone contacts), same decision
// structure as the oracle's gjk_distance.  The simplex lives in 4 fixed register slots (compacted,
// newest vertex last); every slot access is static so nothing spills to scratch memory.
template <typename T> struct Simplex {
    T y[4][3], a[4][3], b[4][3];
    int ia[4], ib[4];          // hull vertex ids of a / b
    int n;
};

template <typename T, typename TV> KS_HD void gjk_support(PairGeo<T, TV>& g, const T* dir, T* y, T* a, T* b) {
#if defined(KS_STAMP_HULL) && defined(__HIP_DEVICE_COMPILE__)
    const long long ts0 = clock64();
#endif
    pair_support(g, dir, T(0), a, b);
#if defined(KS_STAMP_HULL) && defined(__HIP_DEVICE_COMPILE__)
    g.cnt_support += 2;
    g.t_sup += clock64() - ts0;
#endif
    minkowski_point(g, a, b, y);
}

// closest point to the origin on the segment A + t (B - A), t clamped to [0, 1]; returns its squared norm
template <typename T> KS_HD T closest_seg(const T* A, const T* B, T& t) {
    T d[3], q[3];
    sub3(d, B, A);
    const T dd = dot3(d, d), tt = -dot3(A, d);
    t = dd > T(1e-30) ? (tt <= 0 ? T(0) : (tt >= dd ? T(1) : tt / dd)) : T(0);
    copy3(q, A);
    addscl3(q, d, t);
    return dot3(q, q);
}

// Closest point to the origin on a triangle: barycentric weights l[3].  The minimum over four candidates - the closest
// points of the three (clamped) edges and, when the origin projects inside, the interior point - every one of which IS a
// point of the triangle, so the result can only be too far, never outside the simplex.  The textbook region tests
// (Ericson, RTCD 5.1.5; what the fp64 oracle uses) decide on products like d1 d4 - d3 d2 that cancel for the sliver
// triangles a Minkowski difference of two hulls produces (two nearly parallel edges): in fp32 a wrong region gave a
// "closest" point that was not orthogonal to the simplex, GJK then met a repeated support vertex and stopped 3 % short
// of the distance with a normal several degrees off.  No data-dependent branches besides the selects.
template <typename T> KS_HD void closest_tri(const T* A, const T* B, const T* C, T* l) {
    T tab, tac, tbc;
    const T dab = closest_seg(A, B, tab), dac = closest_seg(A, C, tac), dbc = closest_seg(B, C, tbc);
    T best = dab;
    l[0] = 1 - tab; l[1] = tab; l[2] = 0;
    if (dac < best) { best = dac; l[0] = 1 - tac; l[1] = 0; l[2] = tac; }
    if (dbc < best) { best = dbc; l[0] = 0; l[1] = 1 - tbc; l[2] = tbc; }
    T ab[3], ac[3], n[3], c1[3], c2[3];
    sub3(ab, B, A); sub3(ac, C, A);
    cross3(n, ab, ac);
    const T nn = dot3(n, n);
    if (nn > T(1e-30)) {
        // barycentric coordinates of the origin's projection: with ap = -A, l1 = ((ap x ac) . n) / nn, l2 = ((ab x ap) . n) / nn
        cross3(c1, ac, A);
        cross3(c2, A, ab);
        const T l1 = dot3(c1, n) / nn, l2 = dot3(c2, n) / nn, l0 = 1 - l1 - l2;
        if (l0 > 0 && l1 > 0 && l2 > 0) {
            T q[3] = {0, 0, 0};
            addscl3(q, A, l0); addscl3(q, B, l1); addscl3(q, C, l2);
            if (dot3(q, q) < best) { l[0] = l0; l[1] = l1; l[2] = l2; }
        }
    }
}

template <typename T> KS_HD void swap_slots(Simplex<T>& S, T* l, int i, int j, bool doit) {
    // a branch, not selects: the few lanes of a wave that are in a narrow phase rarely need a swap at the same turn, and
    // the wave then skips the ~30 moves (the usual case: a face contact keeps all three vertices)
    if (doit) {
        KS_UNROLL
        for (int c = 0; c < 3; c++) {
            T t;
            t = S.y[i][c]; S.y[i][c] = S.y[j][c]; S.y[j][c] = t;
            t = S.a[i][c]; S.a[i][c] = S.a[j][c]; S.a[j][c] = t;
            t = S.b[i][c]; S.b[i][c] = S.b[j][c]; S.b[j][c] = t;
        }
        const T t = l[i]; l[i] = l[j]; l[j] = t;
        int k;
        k = S.ia[i]; S.ia[i] = S.ia[j]; S.ia[j] = k;
        k = S.ib[i]; S.ib[i] = S.ib[j]; S.ib[j] = k;
    }
}

// closest point on the simplex; reduces it to the supporting sub-simplex (stable compaction),
// lam = weights of the kept vertices, v = closest point.  Returns true if the origin is inside.
template <typename T> KS_HD bool gjk_closest(Simplex<T>& S, T* lam, T* v) {
    T l[4] = {0, 0, 0, 0};
    if (S.n == 1) l[0] = 1;
    else if (S.n == 2) {
        T d[3];
        sub3(d, S.y[1], S.y[0]);
        T t = -dot3(S.y[0], d), dd = dot3(d, d);
        if (t <= 0 || dd < T(1e-15)) { l[0] = 1; l[1] = 0; }
        else if (t >= dd) { l[0] = 0; l[1] = 1; }
        else { l[1] = t / dd; l[0] = 1 - l[1]; }
    } else if (S.n == 3) closest_tri(S.y[0], S.y[1], S.y[2], l);
    else {
        T best = Lim<T>::big;
        bool any = false;
        KS_UNROLL
        for (int f = 0; f < 4; f++) {
            // faces (0,1,2|3) (0,2,3|1) (0,3,1|2) (1,3,2|0)
            const int i0 = f == 3 ? 1 : 0, i1 = f == 0 ? 1 : (f == 1 ? 2 : 3), i2 = f == 0 ? 2 : (f == 1 ? 3 : (f == 2 ? 1 : 2)), i3 = f == 0 ? 3 : (f == 1 ? 1 : (f == 2 ? 2 : 0));
            T ab[3], ac[3], n[3], ad[3];
            sub3(ab, S.y[i1], S.y[i0]); sub3(ac, S.y[i2], S.y[i0]); cross3(n, ab, ac); sub3(ad, S.y[i3], S.y[i0]);
            T sp = -dot3(S.y[i0], n), sd = dot3(ad, n);
            // a sliver tetrahedron cannot certify "inside": evaluate its faces instead
            const bool flat = sd * sd <= T(1e-6) * dot3(n, n) * dot3(ad, ad);
            if (sp * sd < 0 || flat) {
                T lt[3], q[3] = {0, 0, 0};
                closest_tri(S.y[i0], S.y[i1], S.y[i2], lt);
                addscl3(q, S.y[i0], lt[0]); addscl3(q, S.y[i1], lt[1]); addscl3(q, S.y[i2], lt[2]);
                T d2 = dot3(q, q);
                if (d2 < best) {
                    best = d2; any = true;
                    l[0] = l[1] = l[2] = l[3] = 0;
                    l[i0] = lt[0]; l[i1] = lt[1]; l[i2] = lt[2];
                }
            }
        }
        if (!any) return true;
    }
    // stable compaction of the vertices with positive weight (static slot indices only)
    KS_UNROLL
    for (int i = 0; i < 4; i++)
        if (i >= S.n) l[i] = 0;
    KS_UNROLL
    for (int pass = 0; pass < 3; pass++) {
        KS_UNROLL
        for (int i = 0; i < 3; i++) swap_slots(S, l, i, i + 1, !(l[i] > 0) && (l[i + 1] > 0));
    }
    int n = 0;
    KS_UNROLL
    for (int i = 0; i < 4; i++) { lam[i] = l[i]; n += (l[i] > 0) ? 1 : 0; }
    S.n = n;
    v[0] = v[1] = v[2] = 0;
    KS_UNROLL
    for (int i = 0; i < 4; i++)
        if (l[i] > 0) addscl3(v, S.y[i], l[i]);
    return false;
}

// 0: separated by >= margin, 1: contact in the margin zone, 2: overlap (fall back to MPR)
template <typename T> KS_HD void gjk_remember(PairWarm* ws, const Simplex<T>& S) {
    if (ws == nullptr) return;
    if (!packable(S.ia[0], S.ia[1], S.ia[2], S.ib[0], S.ib[1], S.ib[2])) { ws->w[0] = 0; ws->w[1] = 0; return; }
    ws->w[0] = pack3(S.ia[0], S.ia[1], S.ia[2], S.n < 3 ? S.n : 3);
    ws->w[1] = pack3(S.ib[0], S.ib[1], S.ib[2], 0);
}

template <typename T, typename TV> KS_NARROW int gjk_distance(PairGeo<T, TV>& g, T margin, T* dist, T* normal, T* pos, PairWarm* ws = nullptr) {
    Simplex<T> S;
    T lam[4] = {1, 0, 0, 0}, v[3], d[3];
#ifndef KS_GJK_TOL
#define KS_GJK_TOL 1e-6
#endif
    const T tol = T(KS_GJK_TOL);
    // ... plus, in single precision, an absolute floor on the remaining gap |v| - v.w/|v|: the support points are world
    // coordinates of ~0.1 m, so v.w carries ~1e-8 m |v| of rounding and a relative test on distances of 0.01 - 1 mm is decided
    // by noise - the query then crawls across coplanar vertices (a tetrahedron's closest point per turn) for gains of a few
    // nanometres until a vertex repeats.  0.3 um is 3e-4 of the contact margin; the fp64 lane keeps the oracle's test.
#ifndef KS_GJK_GAP
#define KS_GJK_GAP 3e-7
#endif
    const T gap = sizeof(T) == 4 ? T(KS_GJK_GAP) : T(0);
    KS_UNROLL
    for (int i = 0; i < 4; i++) {
        S.ia[i] = 0; S.ib[i] = 0;
        KS_UNROLL
        for (int c = 0; c < 3; c++) { S.y[i][c] = 0; S.a[i][c] = 0; S.b[i][c] = 0; }
    }
    bool warm = false;
    int wia[3] = {0, 0, 0}, wib[3] = {0, 0, 0}, wn = 0, unused = 0;
    if (ws != nullptr) { unpack3(ws->w[0], wia, wn); unpack3(ws->w[1], wib, unused); }
    if (wn > 0) {
        // the previous query's simplex at the current poses
        S.n = wn;
        KS_UNROLL
        for (int i = 0; i < 3; i++) {
            if (i < S.n) {
                S.ia[i] = wia[i]; S.ib[i] = wib[i];
                hull_point(g.R1, g.p1, g.V1, S.ia[i], S.a[i]);
                hull_point(g.R2, g.p2, g.V2, S.ib[i], S.b[i]);
                minkowski_point_ids(g, S.ia[i], S.ib[i], S.a[i], S.b[i], S.y[i]);
            }
        }
        const bool inside = gjk_closest(S, lam, v);
        const T vv0 = dot3(v, v);
        warm = !inside && vv0 == vv0 && vv0 > T(1e-24);             // a degenerate or swallowed simplex -> cold start
        if (warm) { g.hint1 = S.ia[0]; g.hint2 = S.ib[0]; }
    }
    if (!warm) {
        KS_UNROLL
        for (int i = 0; i < 4; i++) {
            KS_UNROLL
            for (int c = 0; c < 3; c++) { S.y[i][c] = 0; S.a[i][c] = 0; S.b[i][c] = 0; }
        }
        lam[0] = 1; lam[1] = 0; lam[2] = 0; lam[3] = 0;
        sub3(d, g.p2, g.p1);
        if (dot3(d, d) < T(1e-15)) { d[0] = 1; d[1] = 0; d[2] = 0; }
        gjk_support(g, d, S.y[0], S.a[0], S.b[0]);
        S.ia[0] = g.hint1; S.ib[0] = g.hint2;
        S.n = 1;
        copy3(v, S.y[0]);
    }
    T last_vw = T(1);
    for (int it = 0; it < 48; it++) {
        T vv = dot3(v, v);
        if (vv < T(1e-24)) { gjk_remember(ws, S); return 2; }
        T nd[3] = {-v[0], -v[1], -v[2]}, w[3], wa[3], wb[3];
        gjk_support(g, nd, w, wa, wb);
        T vw = dot3(v, w);
        last_vw = vw;
#ifdef KS_DEBUG_GJK
        printf("  gjk[%d] it %d n %d vv %.9g vw %.9g |v| %.9g\n", (int)sizeof(T), it, S.n, (double)vv, (double)vw, (double)ksqrt(vv));
#endif
        if (vw > 0 && vw * vw >= margin * margin * vv) { gjk_remember(ws, S); return 0; }
        if (vv - vw <= tol * vv + gap * ksqrt(vv)) break;
        bool dup = false;
        KS_UNROLL
        for (int i = 0; i < 4; i++)
            if (i < S.n && S.y[i][0] == w[0] && S.y[i][1] == w[1] && S.y[i][2] == w[2]) dup = true;
#ifdef KS_DEBUG_GJK
        if (dup) printf("  -> dup\n");
#endif
        if (dup) break;
        Simplex<T> prev = S;
        T plam[4] = {lam[0], lam[1], lam[2], lam[3]}, pv[3] = {v[0], v[1], v[2]};
        KS_UNROLL
        for (int i = 0; i < 4; i++)
            if (i == S.n) { copy3(S.y[i], w); copy3(S.a[i], wa); copy3(S.b[i], wb); S.ia[i] = g.hint1; S.ib[i] = g.hint2; }
        S.n++;
#if defined(KS_STAMP_HULL) && defined(__HIP_DEVICE_COMPILE__)
        const long long tc0 = clock64();
        const bool inside_ = gjk_closest(S, lam, v);
        g.t_clo += clock64() - tc0;
        if (inside_) { gjk_remember(ws, S); return 2; }
#else
        if (gjk_closest(S, lam, v)) { gjk_remember(ws, S); return 2; }
#endif
#ifdef KS_DEBUG_GJK
        printf("  -> new vv %.9g (n %d) lam %.4g %.4g %.4g %.4g%s\n", (double)dot3(v, v), S.n, (double)lam[0], (double)lam[1], (double)lam[2], (double)lam[3], dot3(v, v) >= vv ? " NO DECREASE" : "");
#endif
        if (dot3(v, v) >= vv) {
            copy3(v, pv);
            S = prev;
            lam[0] = plam[0]; lam[1] = plam[1]; lam[2] = plam[2]; lam[3] = plam[3];
            break;
        }
    }
    gjk_remember(ws, S);
    // The iteration ended (no decrease / repeated vertex / tolerance) while the last support point lay BEYOND the origin along -v
    // (v.w < 0): v is then no certified separation - a flat tetrahedron of a near-touching pair can stop here a few um "apart"
    // although the hulls overlap (found on a 64-gon cylinder against a finger: +4.9 um reported, 34 um of penetration).  Such a
    // result is handed to the penetration query first (return 3); only if that finds no overlap does the caller keep it.
#ifndef KS_NO_OPEN_FALLBACK     // (experiment switch)
    const bool open = last_vw < 0;
#else
    const bool open = false;
#endif
    T dd = norm3(v);
    if (dd < T(1e-12)) return 2;
    if (dd >= margin) return open ? 2 : 0;
    T p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0};
    KS_UNROLL
    for (int i = 0; i < 4; i++)
        if (i < S.n) { addscl3(p1, S.a[i], lam[i]); addscl3(p2, S.b[i], lam[i]); }
    *dist = dd;
    KS_UNROLL
    for (int i = 0; i < 3; i++) { normal[i] = -v[i] / dd; pos[i] = T(0.5) * (p1[i] + p2[i]); }
    return open ? 3 : 1;
}


// Exact-result cull for hull pairs: separating-axis test of the two geoms' bounding boxes (half
// extents about the geom origin, in the geom frame).  A gap >= margin along any of the 15 axes
// means the hulls (inside the boxes) are >= margin apart, which is exactly when GJK reports
// "no contact", so skipping the pair never changes the contact set.
template <typename T>
KS_HD bool obb_separated(const T* R1, const T* p1, const T* e1, const T* R2, const T* p2, const T* e2, T margin) {
    T R[3][3], A[3][3], tw[3], t[3];
    sub3(tw, p2, p1);
    mulRtv(t, R1, tw);
    KS_UNROLL
    for (int i = 0; i < 3; i++) {
        KS_UNROLL
        for (int j = 0; j < 3; j++) {
            R[i][j] = R1[i] * R2[j] + R1[3 + i] * R2[3 + j] + R1[6 + i] * R2[6 + j];   // (R1^T R2)[i][j]
            A[i][j] = kabs(R[i][j]) + T(1e-6);
        }
    }
    bool sep = false;
    KS_UNROLL
    for (int i = 0; i < 3; i++) sep = sep || (kabs(t[i]) > e1[i] + e2[0] * A[i][0] + e2[1] * A[i][1] + e2[2] * A[i][2] + margin);
    KS_UNROLL
    for (int j = 0; j < 3; j++)
        sep = sep || (kabs(t[0] * R[0][j] + t[1] * R[1][j] + t[2] * R[2][j]) > e1[0] * A[0][j] + e1[1] * A[1][j] + e1[2] * A[2][j] + e2[j] + margin);
    KS_UNROLL
    for (int i = 0; i < 3; i++) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
        KS_UNROLL
        for (int j = 0; j < 3; j++) {
            const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            T ra = e1[i1] * A[i2][j] + e1[i2] * A[i1][j];
            T rb = e2[j1] * A[i][j2] + e2[j2] * A[i][j1];
            sep = sep || (kabs(t[i2] * R[i1][j] - t[i1] * R[i2][j]) > ra + rb + margin);
        }
    }
    return sep;
}

template <typename T> KS_HD void make_frame(const T* n, T* t1, T* t2) {
    t1[0] = 0; t1[1] = 0; t1[2] = 0;
    if (n[1] < T(0.5) && n[1] > T(-0.5)) t1[1] = 1; else t1[2] = 1;
    T d = dot3(n, t1);
    addscl3(t1, n, -d);
    normalize3(t1);
    cross3(t2, n, t1);
}

#if defined(KS_STAMP) && defined(__HIP_DEVICE_COMPILE__)
#define KS_T0 long long t0_ = clock64();
#define KS_TICK(i) { long long t1_ = clock64(); if (prof) prof[i] += (float)(t1_ - t0_); t0_ = t1_; }
#else
#define KS_T0
#define KS_TICK(i)
#endif

// One staged contact record of pair slot `slot` (record index, see SCR_STAGE)
template <typename T, typename S>
KS_HD void stage_contact(S scr, int slot, int b1, int b2, int pi, T mu, T dist, const T* pos, const T* normal) {
    const int o = SCR_STAGE + slot * STAGE_REC;
    T n[3] = {normal[0], normal[1], normal[2]};
    normalize3(n);
    KS_UNROLL
    for (int i = 0; i < 3; i++) { scr(o + i) = pos[i]; scr(o + 3 + i) = n[i]; }
    scr(o + 6) = dist;
    scr(o + 7) = mu;
    scr(o + 8) = T(b1 + 16 * b2 + 256 * pi);   // pi: the pair, for its margin (explicit pairs 0, dynamic pairs the geoms')
}

// The two culls of a plane pair (bounding sphere, exact box-vs-plane): false = no vertex can be within the margin
template <typename T, typename S>
KS_HD bool plane_may_touch(S scr, KS_LDS const PairRec<T>* prp) {
    KS_LDS const PairRec<T>& pr = *prp;
    const T margin = pr.margin, rbound = pr.rbound2;
    const T size[3] = {pr.size2[0], pr.size2[1], pr.size2[2]};
    const int o = SCR_GP + (pr.g2 - 1) * 12;
    const T ln[3] = {scr(o + 6), scr(o + 7), scr(o + 8)}, cdist = scr(o + 11);
    if (cdist > rbound + margin) return false;
    return !(cdist - (kabs(ln[0]) * size[0] + kabs(ln[1]) * size[1] + kabs(ln[2]) * size[2]) > margin);
}

#if defined(KS_PLANE_HOOK) && !defined(__HIP_DEVICE_COMPILE__)
inline int (*ks_plane_hook)(int g2, const double* R, const double* p, int coarse_best) = nullptr;
#endif
// Ground plane z = 0 (normal +z) vs the hull of geom g2, worked on by the WHOLE team: deepest vertex, then up
// to 3 more within the margin that are > 0.3*rbound from every accepted vertex (index order).  The lanes
// scan the vertex table SUBS rows at a time; the team then agrees on the deepest vertex and on the
// vertices the oracle's greedy index-order rule accepts (see pass 2 below; the 16-lane team walks only the vertices
// its first pass found within the margin).
// Returns the number of contacts staged at record `slot`.
#ifndef KS_PLANE_F64
#define KS_PLANE_F64 2         // fp32 product: 1 = the staged contacts' depths in fp64, 2 = the vertex scans' distances too (selection and margin tests)
#endif
template <typename T, typename S, int SUBS>
KS_HD int collide_plane_hull(S scr, Team<SUBS> team, KS_LDS const PairRec<T>* prp, int slot, float* prof = nullptr) {
    KS_T0
    KS_LDS const PairRec<T>& pr = *prp;
    const T PLANE_MESH_TOL = T(0.3);
    const int g2 = pr.g2, pi = (pr.obj_hand >> 12) & PAIR_INDEX_MASK;
    const T margin = pr.margin, mu = pr.mu, rbound = pr.rbound2;
    const T size[3] = {pr.size2[0], pr.size2[1], pr.size2[2]};
    KS_TAB const T* V = pr.V2;
    // the 16-lane team remembers, per lane, in which of its <= 64 rounds its vertex was within the margin (hulls of <= 1024 vertices:
    // every hull of the standard build, all but the short bottle's base and the lemon in the multi-geom build - decided per pair there);
    // larger hulls are walked a second time instead
    const int nv = pr.n2, body2 = pr.body2;
    const bool MASKED = SUBS == 16 && (HULL_VERT_MAX <= 1024 || nv <= 1024);
    T R2[9], p2[3];
    geom_pose_cached(scr, g2, R2, p2);
    const T cdist = p2[2];
    if (cdist > rbound + margin) return 0;
    const T ln[3] = {R2[6], R2[7], R2[8]};          // R2^T e_z
    // D: the type the vertices' signed distances are formed in, DS: the type of a STAGED contact's depth (KS_PLANE_F64, fp32 product: fp64
    // arithmetic on the fp32 pose and table - the same resting vertex for hundreds of substeps makes the fp32 rounding of `cdist + v.ln`,
    // ~4e-9 m on a depth of 1e-5 .. 1e-4 m, a bias like the one of MPR's read-off, see mpr_penetration)
    using D = std::conditional_t<(sizeof(T) == 4 && KS_PLANE_F64 >= 2), double, T>;
    using DS = std::conditional_t<(sizeof(T) == 4 && KS_PLANE_F64 >= 1), double, T>;
    const D cdistD = (D)cdist, marginD = (D)margin, lnD[3] = {(D)ln[0], (D)ln[1], (D)ln[2]};
    auto pdist = [&](int i) -> D { return cdistD + (D)V[4 * i] * lnD[0] + (D)V[4 * i + 1] * lnD[1] + (D)V[4 * i + 2] * lnD[2]; };
    // exact cull: lowest point of the geom's bounding box (half extents geom_size about the geom origin) is
    // above the margin -> every hull vertex is too
    if (cdist - (kabs(ln[0]) * size[0] + kabs(ln[1]) * size[1] + kabs(ln[2]) * size[2]) > margin) return 0;
    // Pass 1, SUBS consecutive vertices per round (lane k takes vertex base + k: adjacent lanes read adjacent
    // 16-byte rows, no LDS bank conflicts), four rounds of reads in flight: the deepest vertex.
    D bd = D(1e30);
    int best = nv;
    unsigned long long cand = 0;                    // bit r: this lane's vertex of round r (index r SUBS + sub) is within the margin
    unsigned long long near = 0;                    // (fp64 only) bit r: ... was within the tie band of the lane's running minimum when it was seen
    constexpr bool TIE_RULE = sizeof(D) == 8;       // see below
    constexpr D TIE_EPS = D(1e-12);
    for (int base0 = 0; base0 < nv; base0 += 4 * SUBS) {
        D dd[4];
        KS_UNROLL
        for (int u = 0; u < 4; u++) {
            const int i = base0 + u * SUBS + team.sub, ii = i < nv ? i : 0;
            dd[u] = pdist(ii);
        }
        KS_UNROLL
        for (int u = 0; u < 4; u++) {
            const int i = base0 + u * SUBS + team.sub;
            if (MASKED) {
                if (i < nv && dd[u] <= marginD) cand |= 1ull << (base0 / SUBS + u);
                if constexpr (TIE_RULE)
                    if (i < nv && dd[u] <= bd + TIE_EPS) near |= 1ull << (base0 / SUBS + u);
            }
            if (i < nv && dd[u] < bd) { bd = dd[u]; best = i; }
        }
    }
    KS_TICK(12)
    const D bd_lane = bd;
    team.argmin(bd, best);
    if (bd > marginD) return 0;
    // Ties between equally deep vertices (a standing cylinder's rim, a landing cube's four corners) are decided by rounding.  Wherever the
    // distances are formed in fp64 (D = double: the fp64 instantiation - the parity instrument - AND, since round 5, the fp32 product with
    // KS_PLANE_F64 = 2, its default) the oracle's rule applies (ko_physics.c collide_plane_hull): the first contact is the LOWEST-INDEX
    // vertex within 1e-12 m of the deepest one, so that oracle and kernels agree whatever the order of their arithmetic.  Only an fp32 scan
    // (KS_PLANE_F64 < 2) keeps the team's arg-min: the band is below the rounding of fp32 distances.  (A physical dead band of 1 um was
    // measured in round 3 and NOT kept: tools/experiments/r03_plane_tie_rule.patch)
    if constexpr (TIE_RULE) {
        const D lim = bd + TIE_EPS;
        int first = 0x7fffffff;
        if (MASKED) {
            // this lane's vertices that can be within the band of the team's minimum, lowest round first (`near` is a superset)
            unsigned long long c2 = bd_lane <= lim ? near : 0ull;
            while (c2 != 0) {
                const int i = __builtin_ctzll(c2) * SUBS + team.sub;
                if (pdist(i) <= lim) { first = i; break; }
                c2 &= c2 - 1;
            }
            T key = T(0);
            team.argmin(key, first);
        } else {
            // every lane walks its share of the vertices in index order, the team takes the lowest hit
            for (int i = team.sub; i < nv; i += SUBS)
                if (pdist(i) <= lim) { first = i; break; }
            if constexpr (SUBS > 1) {
                T key = T(0);
                team.argmin(key, first);
            }
        }
        if (first != 0x7fffffff) best = first;
    }
#if defined(KS_PLANE_HOOK) && !defined(__HIP_DEVICE_COMPILE__)
    // host-only experiment (tests/studies/divergence_table.py): the first vertex chosen by an external fp64 evaluation on the lane's own pose
    if (ks_plane_hook) { double R[9], pp[3]; for (int k = 0; k < 9; k++) R[k] = (double)R2[k]; for (int k = 0; k < 3; k++) pp[k] = (double)p2[k]; best = ks_plane_hook(g2, R, pp, best); }
#endif
    KS_TICK(4)
    T cv[4][3];
    int nc = 1;
    cv[0][0] = V[4 * best]; cv[0][1] = V[4 * best + 1]; cv[0][2] = V[4 * best + 2];
    T thr2 = PLANE_MESH_TOL * rbound;
    thr2 *= thr2;
    // Pass 2: the oracle's greedy rule is ONE walk over the vertex indices that accepts a vertex when it is within
    // the margin and far from everything accepted before it, so the k-th accepted vertex is the first acceptable
    // index after the (k-1)-th.  The team looks for it SUBS indices at a time (a ballot per round, lowest lane =
    // lowest index) and resumes behind it: at most nv/SUBS + 3 rounds in total, however many vertices lie within
    // the margin (a palm lying flat on the ground has hundreds).
    int start = 0;
    if (MASKED) {
        // The first pass has seen every vertex: each lane kept the rounds in which its vertex was within the margin (nv <=
        // 1024, checked when the model is loaded: 64 rounds).  The walk then only touches those - a finger tip or a cube
        // corner on the ground has a handful, and the old search for a vertex that is not there was a second full scan of the
        // hull (24 rounds for the palm) per plane pair and substep.  Every lane offers its lowest candidate >= start that is far
        // from everything accepted so far (one that is too close stays too close: dropped), the team takes the lowest offer.
        while (nc < 4) {
            int mine = 0x7fffffff;
            while (cand != 0) {
                const int i = __builtin_ctzll(cand) * SUBS + team.sub;
                bool ok = i >= start;
                if (ok) {
                    const T v[3] = {V[4 * i], V[4 * i + 1], V[4 * i + 2]};
                    KS_UNROLL
                    for (int k = 0; k < 3; k++) {
                        if (k < nc) {
                            T dv[3];
                            sub3(dv, v, cv[k]);
                            if (dot3(dv, dv) <= thr2) ok = false;
                        }
                    }
                }
                if (ok) { mine = i; break; }
                cand &= cand - 1;
            }
            T key = T(0);
            int found = mine;
            team.argmin(key, found);
            if (found == 0x7fffffff) break;
            KS_UNROLL
            for (int k = 1; k < 4; k++)
                if (k == nc) { cv[k][0] = V[4 * found]; cv[k][1] = V[4 * found + 1]; cv[k][2] = V[4 * found + 2]; }
            nc++;
            start = found + 1;
        }
    } else
    while (nc < 4 && start < nv) {
        int found = -1;
        for (int base = start; base < nv; base += 2 * SUBS) {
            T v[2][3];
            bool ok[2];
            KS_UNROLL
            for (int u = 0; u < 2; u++) {
                const int i = base + u * SUBS + team.sub, ii = i < nv ? i : 0;
                v[u][0] = V[4 * ii]; v[u][1] = V[4 * ii + 1]; v[u][2] = V[4 * ii + 2];
                ok[u] = i < nv && pdist(ii) <= marginD;
                KS_UNROLL
                for (int k = 0; k < 3; k++) {
                    if (k < nc) {
                        T dv[3];
                        sub3(dv, v[u], cv[k]);
                        if (dot3(dv, dv) <= thr2) ok[u] = false;
                    }
                }
            }
            const unsigned v0 = team.ballot(ok[0]), v1 = team.ballot(ok[1]);
            if (v0) { found = base + kctz(v0); break; }
            if (v1) { found = base + SUBS + kctz(v1); break; }
        }
        if (found < 0) break;
        KS_UNROLL
        for (int k = 1; k < 4; k++)
            if (k == nc) { cv[k][0] = V[4 * found]; cv[k][1] = V[4 * found + 1]; cv[k][2] = V[4 * found + 2]; }
        nc++;
        start = found + 1;
    }
    KS_TICK(23)
    const T normal[3] = {0, 0, 1};
    if constexpr (SUBS >= 4) {
        // lane k stages contact k: every lane holds the four vertices, one instruction stream for all of them
        T c[3] = {cv[0][0], cv[0][1], cv[0][2]};
        KS_UNROLL
        for (int q = 1; q < 4; q++)
            if (team.sub == q) { c[0] = cv[q][0]; c[1] = cv[q][1]; c[2] = cv[q][2]; }
        if (team.sub < nc) {
            const DS dS = (DS)cdist + (DS)c[0] * (DS)ln[0] + (DS)c[1] * (DS)ln[1] + (DS)c[2] * (DS)ln[2];
            T d = (T)dS, w[3];
            mulRv(w, R2, c);
            add3(w, w, p2);
            w[2] -= T(0.5) * d;
            stage_contact(scr, slot + team.sub, 0, body2, pi, mu, d, w, normal);
        }
    } else {
        KS_UNROLL
        for (int k = 0; k < 4; k++) {
            if (k < nc) {
                const DS dS = (DS)cdist + (DS)cv[k][0] * (DS)ln[0] + (DS)cv[k][1] * (DS)ln[1] + (DS)cv[k][2] * (DS)ln[2];
                T d = (T)dS, w[3];
                mulRv(w, R2, cv[k]);
                add3(w, w, p2);
                w[2] -= T(0.5) * d;
                stage_contact(scr, slot + k, 0, body2, pi, mu, d, w, normal);
            }
        }
    }
    return nc;
}

// hull tables, cube maps, poses and support hints of a pair in the operand order of the convex queries (obj_first: see collide_hull_hull)
template <typename T, typename S>
KS_HD void fill_pair_geo(PairGeo<T>& pg, const unsigned short* dirtab, S scr, KS_LDS const PairRec<T>& pr, bool obj_first, int h1, int h2) {
    const int g1 = pr.g1, g2 = pr.g2, flags = pr.obj_hand;
    if (obj_first) {
        pg.V1 = pr.V2; pg.n1 = pr.n2; pg.off1 = pr.off2; pg.adj1 = pr.adj2;
        pg.V2 = pr.V1; pg.n2 = pr.n1; pg.off2 = pr.off1; pg.adj2 = pr.adj1;
        pg.dir1 = dirtab + SUPPORT_CELLS * ((flags >> 8) & 15);
        pg.dir2 = dirtab + SUPPORT_CELLS * ((flags >> 4) & 15);
        geom_pose_cached(scr, g2, pg.R1, pg.p1);
        geom_pose_cached(scr, g1, pg.R2, pg.p2);
    } else {
        pg.V1 = pr.V1; pg.n1 = pr.n1; pg.off1 = pr.off1; pg.adj1 = pr.adj1;
        pg.V2 = pr.V2; pg.n2 = pr.n2; pg.off2 = pr.off2; pg.adj2 = pr.adj2;
        pg.dir1 = dirtab + SUPPORT_CELLS * ((flags >> 4) & 15);
        pg.dir2 = dirtab + SUPPORT_CELLS * ((flags >> 8) & 15);
        geom_pose_cached(scr, g1, pg.R1, pg.p1);
        geom_pose_cached(scr, g2, pg.R2, pg.p2);
    }
    pg.hint1 = h1 < pg.n1 ? h1 : 0;
    pg.hint2 = h2 < pg.n2 ? h2 : 0;
    pg.half_margin = T(0);
#ifdef KS_STAMP_HULL
    pg.cnt_support = 0; pg.cnt_steps = 0; pg.t_sup = 0; pg.t_clo = 0;
#endif
}
// The multi-geom build's fp64 distance query of the fp32 product (KS_MG_GJK_F64, see collide_hull_hull): OUT OF LINE, and it reads the pair's record itself.
// Inlined into `collision` its fp64 simplex takes that function to 256 + 156 registers - past the stepping kernels' budget (ks_api.hip: KS_ROLLOUT_NUM_VGPR),
// the learner's waves no longer fit beside them and its stream falls behind (episodes dropped); with `collision` itself inlined into the kernels the budget
// holds but everything spills (BottleS 3.09 -> 2.46 M env-steps/s); out of line with the caller's pair record live across the call 256 + 134.
template <typename T, typename S>
KS_FN int gjk_distance_f64(const unsigned short* dirtab, S scr, KS_LDS const PairRec<T>* prp, bool obj_first, int& h1, int& h2, T* dist, T* dir, T* pos, PairWarm* ws) {
    PairGeo<double, T> pd;
    {
        PairGeo<T> pg;
        fill_pair_geo(pg, dirtab, scr, *prp, obj_first, h1, h2);
        KS_UNROLL
        for (int k = 0; k < 9; k++) { pd.R1[k] = (double)pg.R1[k]; pd.R2[k] = (double)pg.R2[k]; }
        KS_UNROLL
        for (int k = 0; k < 3; k++) { pd.p1[k] = (double)pg.p1[k]; pd.p2[k] = (double)pg.p2[k]; }
        pd.V1 = pg.V1; pd.V2 = pg.V2; pd.off1 = pg.off1; pd.adj1 = pg.adj1; pd.off2 = pg.off2; pd.adj2 = pg.adj2; pd.dir1 = pg.dir1; pd.dir2 = pg.dir2;
        pd.n1 = pg.n1; pd.n2 = pg.n2; pd.hint1 = pg.hint1; pd.hint2 = pg.hint2; pd.half_margin = 0.0;
#ifdef KS_STAMP_HULL
        pd.cnt_support = 0; pd.cnt_steps = 0; pd.t_sup = 0; pd.t_clo = 0;
#endif
    }
    double dist_d = 0, dir_d[3] = {0, 0, 0}, pos_d[3] = {0, 0, 0};
    const int r = gjk_distance(pd, (double)prp->margin, &dist_d, dir_d, pos_d, ws);
    h1 = pd.hint1; h2 = pd.hint2;
    *dist = (T)dist_d;
    KS_UNROLL
    for (int k = 0; k < 3; k++) { dir[k] = (T)dir_d[k]; pos[k] = (T)pos_d[k]; }
    return r;
}

// hull vs hull (one lane): bounding spheres, exact OBB test, GJK distance for the margin zone, MPR on overlap
// Hull pairs keep the support vertices their last GJK / MPR query ended on as the hill-climbing start of the next
// substep (two 10-bit vertex ids packed above the 3-bit contact count in the pair's SCR_PC word): the climb is
// then a handful of steps instead of a walk across the hull.  The support vertex found does not depend on the start.
constexpr int PC_COUNT_MASK = 7, PC_HINT_BITS = 10, PC_HINT_MAX = (1 << PC_HINT_BITS) - 1;
// hull pairs per lane of a 16-lane team: standard 22 pairs -> 2; multi-geom 15 + 7 + 7 x 8 = 78 -> 5 (capacity NPAIR_MAX - 8 plane pairs at least)
constexpr int HPL = MULTI_GEOM ? 6 : 2;
KS_HD int pc_pack(int count, int h1, int h2) { return count + 8 * (h1 + (1 << PC_HINT_BITS) * h2); }

// The two culls of a hull pair (bounding spheres, exact OBB test): false = the hulls are further apart than the margin
template <typename T, typename S>
KS_HD bool hull_pair_may_touch(S scr, KS_LDS const PairRec<T>* prp) {
    KS_LDS const PairRec<T>& pr = *prp;
    const int g1 = pr.g1, g2 = pr.g2;
    const T margin = pr.margin, bound = pr.rbound1 + pr.rbound2 + margin;
    const T size1[3] = {pr.size1[0], pr.size1[1], pr.size1[2]}, size2[3] = {pr.size2[0], pr.size2[1], pr.size2[2]};
    T R1[9], p1[3], R2[9], p2[3], t[3];
    geom_pose_cached(scr, g1, R1, p1);
    geom_pose_cached(scr, g2, R2, p2);
    sub3(t, p1, p2);
    if (dot3(t, t) > bound * bound) return false;
#ifndef KS_NO_OBB
    if (obb_separated(R1, p1, size1, R2, p2, size2, margin)) return false;
#endif
    return true;
}

// narrow phase of a hull pair that passed hull_pair_may_touch
template <typename T, typename S>
KS_HD int collide_hull_hull(const Model<T>& m, const unsigned short* dirtab, S scr, KS_LDS const PairRec<T>* prp, int slot, int packed_in, int& h1_out,
                            int& h2_out, PairWarm* ws, float* prof = nullptr) {
    KS_LDS const PairRec<T>& pr = *prp;
    PairGeo<T> pg;
    constexpr bool use_sm = (KS_MPR_SM != 0) && sizeof(T) == 4;
#ifdef KS_STAMP_HULL
    const long long th0 = clock64();
#endif
    h1_out = (packed_in >> 3) & PC_HINT_MAX;
    h2_out = (packed_in >> (3 + PC_HINT_BITS)) & PC_HINT_MAX;
    // the whole record first: one burst of LDS reads, one wait
    const int g1 = pr.g1, g2 = pr.g2;
    int body1 = pr.body1, body2 = pr.body2;
    int flags = pr.obj_hand;
    const T margin = pr.margin;
    T mu = (flags & 1) ? T(scr(SCR_ENVP + 1)) : pr.mu;
    // Operand order of the convex queries = MuJoCo's (oracle/ko_physics.c: collide_hull_hull): an explicit <pair geom1="object" geom2=hand geom>
    // reaches libccd with the OBJECT as obj1; the model stores the pair as (hand geom, object).  MPR is not symmetric in its operands
    // (within its 1e-6 tolerance the portal path - the contact point of a pad-on-face contact - depends on the order): rows 35-45 of the
    // recorded MuJoCo 1.50 trajectory agree to 2e-10 / 8e-8 with the object first, to 1.3e-6 with the hand geom first.  For these pairs
    // the queries run on the exchanged operands (hull 1 of `pg` = the object) and the direction is flipped back to g1 -> g2; the
    // hints and the warm words then describe `pg`'s order, consistently from one substep to the next.
#ifndef KS_OBJ_FIRST
#define KS_OBJ_FIRST 1          // 0: the operand order of rounds 1-4 (diagnostic A/B only: tools/r05/operand_order_fp32.py)
#endif
    const bool obj_first = (KS_OBJ_FIRST != 0) && (g2 == OBJ_GEOM);
#ifndef KS_MG_GJK_F64
#define KS_MG_GJK_F64 1         // 0: the fp32 distance query (first half of round 6: 37 of 48 long-horizon envs, 10 - 12 % faster)
#endif
    // Multi-geom build, fp32 product (KS_MG_GJK_F64): the DISTANCE query in fp64 arithmetic on the fp32 poses and the fp32 hull tables - what the fp64
    // instantiation runs.  These objects' pieces collide dynamically in the 1 mm margin zone, where the closest-feature query decides which contact exists; with
    // the penetration query already on fp64 points this is the host study's "hull pairs from an fp64 collision stage" (tools/r06/mg_host_study.py: 44 of 48;
    // on the GPU 43 of 48 through ks_step against 37).  The query runs out of line on its own copy of the pair record (gjk_distance_f64); this function's
    // record is only filled when the penetration query needs it.
    T depth, dist, dir[3], pos[3];
    int r = 2;
#ifdef KS_STAMP_HULL
    const long long th1 = clock64();
#endif
    // (Experiment, off by default.)  A pair without margin (the explicit object pairs) that PENETRATED in the previous substep goes straight
    // to the penetration query (fp32 product only, KS_MPR_FIRST): since round 4 every object contact is a penetration contact, and the distance query in
    // front of it - whose only job for such a pair is to say "overlap" - needs the most iterations of the wave to do so (it must
    // enclose the origin), while the lanes with separated pairs confirm their cached separation in one.  MPR decides overlap itself;
    // the two can only disagree within their tolerances of touching.  The fp64 instantiation keeps the oracle's order.
#ifndef KS_MPR_FIRST
#define KS_MPR_FIRST 0          // measured in round 4 (one box, A/B): sim-only 4.28 M with it, 4.36 M without - the waves run in lockstep, the lanes
#endif                          // that skip the distance query wait for those that do not, and the extra branch costs more than it saves: off
    // (KS_MPR_FIRST=2: EVERY margin-0 pair, penetrating before or not - no distance query at all for the object pairs.  Measured, A/B on one
    // box: training 2.71 M against 3.03 M env-steps/s, sim-only 3.94 against 4.31 M: a separated pair costs the distance query one iteration
    // (its cached separating simplex), the penetration query a cold portal discovery.)
    // Round 6 (KS_MPR_FIRST = 3, with the state-machine query): a margin-0 pair whose previous query got as far as MPR - pair memory word 2 - skips the distance
    // query.  For such a pair MuJoCo itself asks libccd for the penetration only; the distance query in front of it is this repo's shortcut for SEPARATED pairs
    // (a warm confirmation of the cached separating simplex), and for a penetrating pair it is the expensive case (it must enclose the origin).
    const bool mpr_first = (KS_MPR_FIRST == 3) ? (use_sm && ws != nullptr && ws->w[2] != 0u && !(margin > T(0)))
                         : (KS_MPR_FIRST == 2) ? (sizeof(T) == 4 && !(margin > T(0)))
                                               : ((KS_MPR_FIRST != 0) && (KS_MPR_WARM != 0) && sizeof(T) == 4 && ws != nullptr && (ws->w[2] >> 30) == 3u && !(margin > T(0)));
    constexpr bool gjk_f64 = MULTI_GEOM && (KS_MG_GJK_F64 != 0) && sizeof(T) == 4 && (KS_MPR_FIRST == 0);
    if constexpr (gjk_f64) {
        int hh1 = h1_out, hh2 = h2_out;
        r = gjk_distance_f64(dirtab, scr, prp, obj_first, hh1, hh2, &dist, dir, pos, ws);
        fill_pair_geo(pg, dirtab, scr, pr, obj_first, hh1, hh2);        // (this function's own record: for the penetration query, should it come to that)
        // (... and the record's other fields read again rather than held across the call: `collision` must stay inside the stepping kernels' register budget)
        body1 = pr.body1; body2 = pr.body2; flags = pr.obj_hand;
        mu = (flags & 1) ? T(scr(SCR_ENVP + 1)) : pr.mu;
    } else {
        if (obj_first) {
            pg.V1 = pr.V2; pg.n1 = pr.n2; pg.off1 = pr.off2; pg.adj1 = pr.adj2;
            pg.V2 = pr.V1; pg.n2 = pr.n1; pg.off2 = pr.off1; pg.adj2 = pr.adj1;
            pg.dir1 = dirtab + SUPPORT_CELLS * ((flags >> 8) & 15);
            pg.dir2 = dirtab + SUPPORT_CELLS * ((flags >> 4) & 15);
            geom_pose_cached(scr, g2, pg.R1, pg.p1);
            geom_pose_cached(scr, g1, pg.R2, pg.p2);
        } else {
            pg.V1 = pr.V1; pg.n1 = pr.n1; pg.off1 = pr.off1; pg.adj1 = pr.adj1;
            pg.V2 = pr.V2; pg.n2 = pr.n2; pg.off2 = pr.off2; pg.adj2 = pr.adj2;
            pg.dir1 = dirtab + SUPPORT_CELLS * ((flags >> 4) & 15);
            pg.dir2 = dirtab + SUPPORT_CELLS * ((flags >> 8) & 15);
            geom_pose_cached(scr, g1, pg.R1, pg.p1);
            geom_pose_cached(scr, g2, pg.R2, pg.p2);
        }
        // hints are only meaningful when they index the pair's own hulls (always, unless a hull has > 1024 vertices)
        pg.hint1 = h1_out < pg.n1 ? h1_out : 0;
        pg.hint2 = h2_out < pg.n2 ? h2_out : 0;
        pg.half_margin = T(0);
#ifdef KS_STAMP_HULL
        pg.cnt_support = 0; pg.cnt_steps = 0; pg.t_sup = 0; pg.t_clo = 0;
#endif
        if (!mpr_first) r = gjk_distance(pg, margin, &dist, dir, pos, ws);
    }
#ifdef KS_STAMP_HULL
    if (prof) { prof[24] += 1.f; prof[25] += (float)pg.cnt_support; prof[27] += (float)(clock64() - th1); prof[28] += (float)pg.t_sup; prof[29] += (float)pg.t_clo; }
#endif
    h1_out = pg.hint1 <= PC_HINT_MAX ? pg.hint1 : 0; h2_out = pg.hint2 <= PC_HINT_MAX ? pg.hint2 : 0;   // (a vertex id beyond the 10-bit field is not remembered: start 0, not a masked id)
    const int pi = (flags >> 12) & PAIR_INDEX_MASK;
    if (obj_first && (r == 1 || r == 3)) { dir[0] = -dir[0]; dir[1] = -dir[1]; dir[2] = -dir[2]; }
    if (use_sm && r < 2 && ws != nullptr) ws->w[2] = 0u;
    if (r == 1) { stage_contact(scr, slot, body1, body2, pi, mu, dist, pos, dir); return 1; }
    if (r >= 2) {
        // 2: overlap (or undecided beyond the margin), 3: a margin-zone result that is not a certified separation
        T mdir[3], mpos[3];
        bool hit;
#ifdef KS_STAMP_HULL
        const long long tm0 = clock64(), clo0 = pg.t_clo;
        const int sup0 = pg.cnt_support;
#endif
        if constexpr (use_sm) {
            hit = mpr_penetration_sm(pg, m.mpr_tol, m.mpr_iters, &depth, mdir, mpos);
            if (ws != nullptr) ws->w[2] = 1u;                   // (pair memory word 2: the pair's last query got as far as MPR - KS_MPR_FIRST=3)
        } else {
            hit = mpr_penetration(pg, m.mpr_tol, m.mpr_iters, &depth, mdir, mpos, ws);
        }
#ifdef KS_STAMP_HULL
        if (prof) { prof[7] += (float)(clock64() - tm0); prof[22] += (float)(pg.cnt_support - sup0) * 0.5f; prof[26] += 1.f; prof[29] += (float)(pg.t_clo - clo0); }     // MPR cycles, support pairs, queries, cycles inside its supports (diagnostic)
#endif
        h1_out = pg.hint1 <= PC_HINT_MAX ? pg.hint1 : 0; h2_out = pg.hint2 <= PC_HINT_MAX ? pg.hint2 : 0;   // (a vertex id beyond the 10-bit field is not remembered: start 0, not a masked id)
        if (hit) {
            if (obj_first) { mdir[0] = -mdir[0]; mdir[1] = -mdir[1]; mdir[2] = -mdir[2]; }
            stage_contact(scr, slot, body1, body2, pi, mu, -depth, mpos, mdir);
            return 1;
        }
        if (r == 3) { stage_contact(scr, slot, body1, body2, pi, mu, dist, pos, dir); return 1; }
        if (use_sm && mpr_first && ws != nullptr) ws->w[2] = 0u;        // no longer penetrating: the next query starts with the distance query again
    }
    return 0;
}

// ---- collide_hull_hull for the two-lane penetration query (mpr_penetration_pair): entered by EVERY lane of the wave; `have` says whether the
// lane has a live pair in this pass.  Culls and the distance query as above, on the owner alone; then the lanes whose pair overlaps take the team
// lane 8 places away as their helper - when both lanes of such a couple have an overlapping pair the lower one goes first and helps the other
// afterwards (a second turn of the loop below, rare: the pairs that overlap together - object against finger links - sit on neighbouring lanes).
static_assert(NCON_MAX * CON_STRIDE >= 16 * 24, "the contact region holds a 24-word portal per lane of a team");
#ifndef KS_MPR_SPLIT
#ifdef KS_MULTI_GEOM
#define KS_MPR_SPLIT 0          // the multi-geom build keeps the one-lane query: measured with the two-lane one BottleS 3.09 -> 3.02 M, BowlS 2.07 -> 2.03 M env-steps/s, the 14-key
#else                           //  stage context 0.545 -> 0.543 M (up to six passes per substep that every lane of the wave now takes, hull tables in global memory)
#define KS_MPR_SPLIT 1          // 0: every penetration query on its owner's lane alone (mpr_penetration_sm), the A/B of the two-lane query
#endif
#endif
#if defined(__HIP_DEVICE_COMPILE__)
template <typename T, typename S, int SUBS>
__device__ __forceinline__ int collide_hull_hull_split(const Model<T>& m, const unsigned short* dirtab, S scr, Team<SUBS> team, bool have, KS_LDS const PairRec<T>* prp,
                                                       int slot, int packed_in, int& h1_out, int& h2_out, PairWarm* ws, float* prof = nullptr) {
    static_assert(SUBS == 16 && KS_MPR_FIRST == 0, "the two-lane query: 16-lane teams, the distance query first");
#ifdef KS_STAMP_SPLIT
    const long long ts0_ = clock64();     // diagnostic build (tools/r06/split_stamp.py): wave-level cycles of the distance-query part and of the penetration part of a pass
#endif
    KS_LDS const PairRec<T>& pr = *prp;
    PairGeo<T> pg;
    int r = 0, c = 0;
    bool obj_first = false;
    T dist = 0, dir[3] = {0, 0, 0}, pos[3] = {0, 0, 0};
    h1_out = (packed_in >> 3) & PC_HINT_MAX;
    h2_out = (packed_in >> (3 + PC_HINT_BITS)) & PC_HINT_MAX;
    if (have) {
        const int g1 = pr.g1, g2 = pr.g2;
        const int flags = pr.obj_hand;
        const T margin = pr.margin;
        obj_first = (KS_OBJ_FIRST != 0) && (g2 == OBJ_GEOM);       // (operand order: see collide_hull_hull)
        if (obj_first) {
            pg.V1 = pr.V2; pg.n1 = pr.n2; pg.off1 = pr.off2; pg.adj1 = pr.adj2;
            pg.V2 = pr.V1; pg.n2 = pr.n1; pg.off2 = pr.off1; pg.adj2 = pr.adj1;
            pg.dir1 = dirtab + SUPPORT_CELLS * ((flags >> 8) & 15);
            pg.dir2 = dirtab + SUPPORT_CELLS * ((flags >> 4) & 15);
            geom_pose_cached(scr, g2, pg.R1, pg.p1);
            geom_pose_cached(scr, g1, pg.R2, pg.p2);
        } else {
            pg.V1 = pr.V1; pg.n1 = pr.n1; pg.off1 = pr.off1; pg.adj1 = pr.adj1;
            pg.V2 = pr.V2; pg.n2 = pr.n2; pg.off2 = pr.off2; pg.adj2 = pr.adj2;
            pg.dir1 = dirtab + SUPPORT_CELLS * ((flags >> 4) & 15);
            pg.dir2 = dirtab + SUPPORT_CELLS * ((flags >> 8) & 15);
            geom_pose_cached(scr, g1, pg.R1, pg.p1);
            geom_pose_cached(scr, g2, pg.R2, pg.p2);
        }
        pg.hint1 = h1_out < pg.n1 ? h1_out : 0;
        pg.hint2 = h2_out < pg.n2 ? h2_out : 0;
        pg.half_margin = T(0);
        r = gjk_distance(pg, margin, &dist, dir, pos, ws);
        h1_out = pg.hint1 <= PC_HINT_MAX ? pg.hint1 : 0; h2_out = pg.hint2 <= PC_HINT_MAX ? pg.hint2 : 0;
        if (obj_first && (r == 1 || r == 3)) { dir[0] = -dir[0]; dir[1] = -dir[1]; dir[2] = -dir[2]; }
        if (r < 2 && ws != nullptr) ws->w[2] = 0u;
        if (r == 1) {
            const T mu = (flags & 1) ? T(scr(SCR_ENVP + 1)) : pr.mu;
            stage_contact(scr, slot, pr.body1, pr.body2, (flags >> 12) & PAIR_INDEX_MASK, mu, dist, pos, dir);
            c = 1;
        }
    }
    // the penetration queries of this pass: owner + helper
    bool pending = have && r >= 2, hit = false;
#ifdef KS_STAMP_SPLIT
    const long long ts1_ = clock64();
    if (prof) { prof[27] += (float)(ts1_ - ts0_); prof[24] += have ? 1.f : 0.f; prof[25] += pending ? 1.f : 0.f; prof[26] += __any(pending) ? 1.f : 0.f; }
#endif
    T depth = 0, mdir[3] = {0, 0, 0}, mpos[3] = {0, 0, 0};
#pragma clang loop unroll(disable)
    while (__any(pending)) {
        const bool partner_pending = dpp_partner(pending ? 1 : 0) != 0;
        const bool own = pending && (!partner_pending || (team.sub & 8) == 0);
        const bool help = dpp_partner(own ? 1 : 0) != 0;
        if (own || help) {
            int hh1 = 0, hh2 = 0;
            T dd = 0, md[3] = {0, 0, 0}, mp[3] = {0, 0, 0};
#ifdef KS_SPLIT_USE_SM
            // diagnostic build: the split caller around the ONE-lane query (what the change of the caller alone does to the bits)
            bool ht = false;
            if (own) { PairGeo<T> pc = pg; ht = mpr_penetration_sm(pc, m.mpr_tol, m.mpr_iters, &dd, md, mp); hh1 = pc.hint1; hh2 = pc.hint2; }
#else
#ifdef KS_STAMP_SPLIT
            int turns_ = 0;
            const bool ht = mpr_penetration_pair(pg, own, (KS_LDS double*)(scr.base + SCR_CON + 24 * team.sub), m.mpr_tol, m.mpr_iters, hh1, hh2, &dd, md, mp, &turns_);
            if (prof && own) prof[29] += (float)turns_;
#else
            const bool ht = mpr_penetration_pair(pg, own, (KS_LDS double*)(scr.base + SCR_CON + 24 * team.sub), m.mpr_tol, m.mpr_iters, hh1, hh2, &dd, md, mp);
#endif
#endif
#ifdef KS_SPLIT_CHECK
            if (own) {      // diagnostic build: the one-lane query on the same pair record must return the same bits
                PairGeo<T> pc = pg;
                T d1 = 0, m1[3] = {0, 0, 0}, p1[3] = {0, 0, 0};
                const bool h1 = mpr_penetration_sm(pc, m.mpr_tol, m.mpr_iters, &d1, m1, p1);
                if (h1 != ht || (ht && (d1 != dd || m1[0] != md[0] || m1[1] != md[1] || m1[2] != md[2] || p1[0] != mp[0] || p1[1] != mp[1] || p1[2] != mp[2])) || pc.hint1 != hh1 || pc.hint2 != hh2)
                    printf("split mismatch lane %d: hit %d/%d depth %.9g/%.9g dir %.9g %.9g %.9g / %.9g %.9g %.9g pos %.9g %.9g %.9g / %.9g %.9g %.9g hints %d %d / %d %d\n", (int)threadIdx.x, (int)h1, (int)ht, (double)d1, (double)dd,
                           (double)m1[0], (double)m1[1], (double)m1[2], (double)md[0], (double)md[1], (double)md[2], (double)p1[0], (double)p1[1], (double)p1[2], (double)mp[0], (double)mp[1], (double)mp[2], pc.hint1, pc.hint2, hh1, hh2);
            }
#endif
            if (own) {
                hit = ht; depth = dd;
                KS_UNROLL
                for (int k = 0; k < 3; k++) { mdir[k] = md[k]; mpos[k] = mp[k]; }
                pg.hint1 = hh1; pg.hint2 = hh2;
                pending = false;
            }
        }
    }
#ifdef KS_STAMP_SPLIT
    if (prof) prof[28] += (float)(clock64() - ts1_);
#endif
    if (have && r >= 2) {
        if (ws != nullptr) ws->w[2] = 1u;
        h1_out = pg.hint1 <= PC_HINT_MAX ? pg.hint1 : 0; h2_out = pg.hint2 <= PC_HINT_MAX ? pg.hint2 : 0;
        const int flags = pr.obj_hand;
        const T mu = (flags & 1) ? T(scr(SCR_ENVP + 1)) : pr.mu;
        const int pi = (flags >> 12) & PAIR_INDEX_MASK;
        if (hit) {
            if (obj_first) { mdir[0] = -mdir[0]; mdir[1] = -mdir[1]; mdir[2] = -mdir[2]; }
            stage_contact(scr, slot, pr.body1, pr.body2, pi, mu, -depth, mpos, mdir);
            c = 1;
        } else if (r == 3) {
            stage_contact(scr, slot, pr.body1, pr.body2, pi, mu, dist, pos, dir);
            c = 1;
        }
    }
    return c;
}
#endif

// the per-pair words (contact count + support hints) must start from zero once per launch: the hints of a fresh
// LDS block are garbage
template <typename T, typename S, int SUBS> KS_HD void reset_pair_words(S scr, Team<SUBS> team) {
    for (int k = team.sub; k < NPAIR_MAX; k += SUBS) scr(SCR_PC + k) = T(0);
    team.sync();
}

template <typename T, typename S, int SUBS>
KS_FN_COLLISION void collision(const Model<T>& m, const Hulls<T>& hu, S scr, Team<SUBS> team, int& ncon, int& status, PairWarm* warm = nullptr,
                     float* prof = nullptr) {
    KS_T0
    const int npair = hu.npair, nhull = hu.nhull;
    KS_LDS const PairRec<T>* pairs = hu.pair;
    const unsigned short* dirtab = m.mesh_dirtab;
    // plane pairs, culls: one pair per lane; the survivors (typically just the object) are then scanned by the
    // whole team, one pair at a time (team-uniform control flow)
    const int nplane = hu.nplane;
    unsigned live = 0;
    for (int k0 = 0; k0 < nplane; k0 += SUBS) {
        const int k = k0 + team.sub;
        bool pass = false;
        if (k < nplane) {
            const int pi = hu.plane_pi[k];
            if (DYNAMIC_SLOTS || pairs[pi].slot + 4 <= NSTAGE) pass = plane_may_touch(scr, pairs + pi);
            else status |= ST_CONTACT_OVERFLOW;
            if (!pass) scr(SCR_PC + pi) = T(0);
        }
        live |= team.ballot(pass) << k0;
    }
    KS_TICK(11)
    int next_slot = 0;                                  // (dynamic slots) first free staging record, team-uniform
    for (unsigned mk = live; mk != 0; mk &= mk - 1) {
        const int pi = hu.plane_pi[kctz(mk)];
        int slot = pairs[pi].slot;
        if constexpr (DYNAMIC_SLOTS) {
            slot = next_slot;
            if (slot + 4 > NSTAGE) { status |= ST_CONTACT_OVERFLOW; if (team.sub == 0) scr(SCR_PC + pi) = T(0); continue; }
            next_slot += 4;
            if (team.sub == 0) scr(SCR_SLOT + pi) = T(slot);
        }
        const int c = collide_plane_hull(scr, team, pairs + pi, slot, prof);
        if (team.sub == 0) scr(SCR_PC + pi) = T(c);
    }
    KS_TICK(8)
    // Hull pairs.  Every lane owns up to HPL of them - round 0: pair `sub`; round r >= 1: the pairs r SUBS .. of the list, dealt to
    // the LAST lanes when the round is not full (standard build: two rounds, the nhull - SUBS pairs beyond the first SUBS) - culls
    // them all, and then runs the narrow phase in as many passes as the busiest lane of the wave has live pairs.  A narrow phase is a
    // few thousand dependent instructions and the wave waits for its slowest lane, so what matters is the number of passes that have
    // any work in them: a second pass is empty unless one lane of the wave has two live pairs at once (round-robin dealing ran
    // two full passes whenever a pair of the second round was live anywhere in the wave).  nhull <= HPL SUBS is checked
    // when the model is loaded.
    if constexpr (SUBS == 1) {
        for (int hk = 0; hk < nhull; hk++) {
            const int pi = hu.hull_pi[hk], word = (int)scr(SCR_PC + pi);
            int c = 0, h1 = (word >> 3) & PC_HINT_MAX, h2 = (word >> (3 + PC_HINT_BITS)) & PC_HINT_MAX;
            if constexpr (DYNAMIC_SLOTS) {
                if (hull_pair_may_touch(scr, pairs + pi)) {
                    if (next_slot + 1 > NSTAGE) status |= ST_CONTACT_OVERFLOW;
                    else {
                        scr(SCR_SLOT + pi) = T(next_slot);
                        c = collide_hull_hull(m, dirtab, scr, pairs + pi, next_slot, word, h1, h2, warm ? warm + hk : nullptr, prof);
                        next_slot++;
                    }
                }
            } else {
                if (pairs[pi].slot + 1 > NSTAGE) status |= ST_CONTACT_OVERFLOW;
                else if (hull_pair_may_touch(scr, pairs + pi)) c = collide_hull_hull(m, dirtab, scr, pairs + pi, pairs[pi].slot, word, h1, h2, warm ? warm + hk : nullptr, prof);
            }
            scr(SCR_PC + pi) = T(pc_pack(c, h1, h2));
        }
    } else {
        int pi_[HPL], word_[HPL], slot_[HPL];
        unsigned todo = 0;
        KS_UNROLL
        for (int r = 0; r < HPL; r++) {
            const int left = nhull - r * SUBS, cnt = left < SUBS ? left : SUBS;           // pairs of this round (<= 0: none)
            const bool have = r == 0 ? team.sub < cnt : team.sub >= SUBS - cnt;
            pi_[r] = 0; word_[r] = 0; slot_[r] = 0;
            bool live_r = false;
            if (have) {
                pi_[r] = hu.hull_pi[r == 0 ? team.sub : r * SUBS + team.sub - (SUBS - cnt)];
                word_[r] = (int)scr(SCR_PC + pi_[r]);
                if (DYNAMIC_SLOTS || pairs[pi_[r]].slot + 1 <= NSTAGE) live_r = hull_pair_may_touch(scr, pairs + pi_[r]);
                else status |= ST_CONTACT_OVERFLOW;
            }
            if constexpr (DYNAMIC_SLOTS) {
                // the pairs that passed the culls take the next free staging records, in (round, lane) order
                if (left > 0) {                                                         // (team-uniform)
                    int taken;
                    slot_[r] = next_slot + team.scan(live_r ? 1 : 0, taken);
                    next_slot += taken;
                    if (live_r && slot_[r] + 1 > NSTAGE) { live_r = false; status |= ST_CONTACT_OVERFLOW; }
                    if (live_r) scr(SCR_SLOT + pi_[r]) = T(slot_[r]);
                }
            }
            if (have) {
                if (!live_r) scr(SCR_PC + pi_[r]) = T(word_[r] & ~PC_COUNT_MASK);     // no contact, hints kept
                todo |= live_r ? (1u << r) : 0u;
            }
        }
#if defined(KS_STAMP) && defined(__HIP_DEVICE_COMPILE__) && !defined(KS_STAMP_SPLIT)
        if (prof) {     // diagnostic build: live hull pairs of this lane / of the busiest lane of the wave (= narrow-phase passes of the wave) / of the team
            const int np = __builtin_popcount(todo);
            int wm = np;
            for (int msk = 32; msk >= 1; msk >>= 1) { const int o = __shfl_xor(wm, msk); wm = o > wm ? o : wm; }
            prof[24] += (float)np; prof[25] += (float)wm; prof[26] += (float)__builtin_popcount(team.ballot(np > 0));
            prof[27] += (float)(todo & 1u); prof[28] += (float)((todo >> 1) & 1u);      // this lane's pair of round 0 / round 1 live
        }
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KS_STAMP_HULL)
        constexpr bool split_queries = (KS_MPR_SPLIT != 0) && (KS_MPR_SM != 0) && sizeof(T) == 4 && SUBS == 16 && KS_MPR_FIRST == 0;
#else
        constexpr bool split_queries = false;
#endif
        if constexpr (split_queries) {
#if defined(__HIP_DEVICE_COMPILE__)
            // the two-lane penetration query: every lane of the wave takes every turn (a lane without a live pair may be its partner's helper)
            while (__any(todo != 0)) {
                const bool have = todo != 0;
                int r = 0, pi_r = pi_[0], word_r = word_[0], slot_r = slot_[0];
                if constexpr (HPL == 2) { r = (todo & 1u) ? 0 : 1; pi_r = r ? pi_[1] : pi_[0]; word_r = r ? word_[1] : word_[0]; slot_r = r ? slot_[1] : slot_[0]; }
                else {
                    r = have ? kctz(todo) : 0;
                    KS_UNROLL
                    for (int q = 1; q < HPL; q++)
                        if (q == r) { pi_r = pi_[q]; word_r = word_[q]; slot_r = slot_[q]; }
                }
                int h1 = 0, h2 = 0;
                const int c = collide_hull_hull_split(m, dirtab, scr, team, have, pairs + pi_r, DYNAMIC_SLOTS ? slot_r : pairs[pi_r].slot, word_r, h1, h2, warm ? warm + r : nullptr, prof);
                if (have) {
                    scr(SCR_PC + pi_r) = T(pc_pack(c, h1, h2));
                    todo &= todo - 1;
                }
            }
#endif
        } else
        // one call site: as many turns as the busiest lane of the wave has live pairs
        while (todo != 0) {
            int r = 0, pi_r = pi_[0], word_r = word_[0], slot_r = slot_[0];
            if constexpr (HPL == 2) { r = (todo & 1u) ? 0 : 1; pi_r = r ? pi_[1] : pi_[0]; word_r = r ? word_[1] : word_[0]; slot_r = r ? slot_[1] : slot_[0]; }
            else {
                r = kctz(todo);
                if constexpr (MULTI_GEOM && (KS_MG_GJK_F64 != 0) && sizeof(T) == 4) {
                    // (the pair of round r re-derived from the lists in LDS instead of kept in 3 x HPL registers across the out-of-line distance query:
                    //  what `collision` holds across that call decides whether the stepping kernels stay inside their register budget)
                    const int left = nhull - r * SUBS, cnt = left < SUBS ? left : SUBS;
                    pi_r = hu.hull_pi[r == 0 ? team.sub : r * SUBS + team.sub - (SUBS - cnt)];
                    word_r = (int)scr(SCR_PC + pi_r);
                    slot_r = DYNAMIC_SLOTS ? (int)scr(SCR_SLOT + pi_r) : 0;
                } else {
                    KS_UNROLL
                    for (int q = 1; q < HPL; q++)
                        if (q == r) { pi_r = pi_[q]; word_r = word_[q]; slot_r = slot_[q]; }
                }
            }
            int h1 = 0, h2 = 0;
            const int c = collide_hull_hull(m, dirtab, scr, pairs + pi_r, DYNAMIC_SLOTS ? slot_r : pairs[pi_r].slot, word_r, h1, h2, warm ? warm + r : nullptr, prof);
            scr(SCR_PC + pi_r) = T(pc_pack(c, h1, h2));
            todo &= todo - 1;
        }
    }
    KS_TICK(9)
    team.sync();
    // merge the staged records in pair order (= the oracle's contact order); contacts beyond NCON_MAX are dropped.
    // Every lane copies the records of the pairs pi = sub (mod SUBS).
    int total = 0;
    int before[(NPAIR_MAX + SUBS - 1) / SUBS];
    if constexpr (SUBS == 16) {
        // lane `sub` owns pairs sub, sub + 16, ..: a row scan per round gives every pair the number of contacts before it
        KS_UNROLL
        for (int q = 0; q < (NPAIR_MAX + SUBS - 1) / SUBS; q++) {
            const int cq = team.sub + q * SUBS < npair ? ((int)scr(SCR_PC + team.sub + q * SUBS) & PC_COUNT_MASK) : 0;
            int tq;
            before[q] = total + team.scan(cq, tq);
            total += tq;
        }
    } else {
        T cnt[NPAIR_MAX];
        KS_UNROLL
        for (int j = 0; j < NPAIR_MAX; j++) cnt[j] = scr(SCR_PC + j);
        KS_UNROLL
        for (int q = 0; q < (NPAIR_MAX + SUBS - 1) / SUBS; q++) before[q] = 0;
        KS_UNROLL
        for (int j = 0; j < NPAIR_MAX; j++) {
            const int c = j < npair ? ((int)cnt[j] & PC_COUNT_MASK) : 0;
            KS_UNROLL
            for (int q = 0; q < (NPAIR_MAX + SUBS - 1) / SUBS; q++) before[q] += (j < team.sub + q * SUBS) ? c : 0;
            total += c;
        }
    }
    KS_UNROLL
    for (int q = 0; q < (NPAIR_MAX + SUBS - 1) / SUBS; q++) {
        const int pi = team.sub + q * SUBS;
        if (pi < npair) {
            const int c = (int)scr(SCR_PC + pi) & PC_COUNT_MASK;
            const int slot = (DYNAMIC_SLOTS && c > 0) ? (int)scr(SCR_SLOT + pi) : pairs[pi].slot;
            for (int k = 0; k < c; k++) {
                const int src = SCR_STAGE + (slot + k) * STAGE_REC, dst = SCR_CON + (before[q] + k) * CON_STRIDE;
                if (before[q] + k < NCON_MAX) {
                    KS_UNROLL
                    for (int f = 0; f < 9; f++) scr(dst + f) = scr(src + f);
                }
            }
        }
    }
    if (total > NCON_MAX) status |= ST_CONTACT_OVERFLOW;
    ncon = total < NCON_MAX ? total : NCON_MAX;
    team.sync();
    KS_TICK(10)
}

// ---------------------------------------------------------------- S5 constraint rows
template <typename T> KS_HD T impedance(const T* solimp, T x) {
    T dmin = solimp[0], dmax = solimp[1], width = solimp[2], y;
    x = kabs(x) / width;
    if (x >= 1) return dmax;
    if (x <= T(0.5)) y = 2 * x * x; else y = 1 - 2 * (1 - x) * (1 - x);
    return dmin + y * (dmax - dmin);
}

// Basis Jacobian of a contact, B[a][j] = frame_a . (Jp_b2(pos) - Jp_b1(pos))[:,j], rebuilt from
// the contact geometry and the body poses in scratch (nothing per-contact is stored but 9 floats).
template <typename T, typename S>
KS_HD void contact_basis(S scr, int ci, T B[3][NV], T& dist, T& mu) {
    const int o = SCR_CON + ci * CON_STRIDE;
    T pos[3] = {scr(o), scr(o + 1), scr(o + 2)};
    T fr[3][3];
    fr[0][0] = scr(o + 3); fr[0][1] = scr(o + 4); fr[0][2] = scr(o + 5);
    make_frame(fr[0], fr[1], fr[2]);
    dist = scr(o + 6);
    mu = scr(o + 7);
    const int bb = (int)scr(o + 8);
    T Jd[3][NV];
    KS_UNROLL
    for (int i = 0; i < 3; i++) {
        KS_UNROLL
        for (int j = 0; j < NV; j++) Jd[i][j] = 0;
    }
    // kinematic data comes from the body poses in LDS (one ds_read each), not from the caller's stack
    KS_UNROLL
    for (int side = 0; side < 2; side++) {
        const int b = side == 0 ? (bb & 15) : ((bb >> 4) & 15);
        const T sg = side == 0 ? T(-1) : T(1);
        if (b >= 2 && b <= 8) {
            KS_UNROLL
            for (int s = 0; s < 3; s++) {
                KS_UNROLL
                for (int i = 0; i < 3; i++) Jd[i][s] += sg * scr(SCR_AX + 3 * s + i);
            }
        }
        if (b >= 3 && b <= 8) {
            const int f = (b - 3) >> 1;                       // finger index
            const int oP = SCR_BP + (1 + 2 * f) * 12, oD = oP + 12;
            T z[3] = {scr(oP + 2), scr(oP + 5), scr(oP + 8)}, r[3], c[3], pp[3] = {scr(oP + 9), scr(oP + 10), scr(oP + 11)};
            sub3(r, pos, pp);
            cross3(c, z, r);
            KS_UNROLL
            for (int ff = 0; ff < 3; ff++) {
                if (ff == f) {
                    KS_UNROLL
                    for (int i = 0; i < 3; i++) Jd[i][3 + 2 * ff] += sg * c[i];
                }
            }
            if ((b - 3) & 1) {
                T pd[3] = {scr(oD + 9), scr(oD + 10), scr(oD + 11)};
                sub3(r, pos, pd);
                cross3(c, z, r);
                KS_UNROLL
                for (int ff = 0; ff < 3; ff++) {
                    if (ff == f) {
                        KS_UNROLL
                        for (int i = 0; i < 3; i++) Jd[i][4 + 2 * ff] += sg * c[i];
                    }
                }
            }
        }
        if (b == 9) {
            const int oO = SCR_BP + 7 * 12;
            T r[3], po[3] = {scr(oO + 9), scr(oO + 10), scr(oO + 11)};
            sub3(r, pos, po);
            KS_UNROLL
            for (int a = 0; a < 3; a++) {
                Jd[a][9 + a] += sg;
                T axv[3] = {scr(oO + a), scr(oO + 3 + a), scr(oO + 6 + a)}, c[3];
                cross3(c, axv, r);
                KS_UNROLL
                for (int i = 0; i < 3; i++) Jd[i][12 + a] += sg * c[i];
            }
        }
    }
    KS_UNROLL
    for (int a = 0; a < 3; a++) {
        KS_UNROLL
        for (int j = 0; j < NV; j++) B[a][j] = fr[a][0] * Jd[0][j] + fr[a][1] * Jd[1][j] + fr[a][2] * Jd[2][j];
    }
}

// out-of-line rebuild for the (rare) contacts beyond the cache: keeps the solver's code size down
template <typename T, typename S>
KS_FN void contact_basis_rebuild(S scr, int ci, T B[3][NV], T& dist, T& mu) { contact_basis<T>(scr, ci, B, dist, mu); }

// basis of contact ci: from the LDS cache when it has a slot (filled by make_constraints), else rebuilt
template <typename T, typename S>
KS_HD void contact_basis_cached(S scr, int ci, T B[3][NV], T& dist, T& mu) {
    if (ci < NBCACHE) {
        const int o = SCR_CON + ci * CON_STRIDE, c = SCR_BCACHE + ci * BC_STRIDE;
        dist = scr(o + 6);
        mu = scr(o + 7);
        KS_UNROLL
        for (int a = 0; a < 3; a++) {
            KS_UNROLL
            for (int j = 0; j < NV; j++) B[a][j] = scr(c + a * NV + j);
        }
    } else {
        // only the temporary escapes to the out-of-line call: B itself stays promotable to registers
        T tmp[3][NV];
        contact_basis_rebuild<T>(scr, ci, tmp, dist, mu);
        KS_UNROLL
        for (int a = 0; a < 3; a++) {
            KS_UNROLL
            for (int j = 0; j < NV; j++) B[a][j] = tmp[a][j];
        }
    }
}

// column i (3 values: normal, tangent1, tangent2 rows) of the basis Jacobian of contact ci
template <typename T, typename S>
KS_HD void contact_basis_column(S scr, int ci, int i, T b[3]) {
    if (ci < NBCACHE) {
        const int c = SCR_BCACHE + ci * BC_STRIDE;
        const int ii = i < NV ? i : 0;
        KS_UNROLL
        for (int a = 0; a < 3; a++) b[a] = i < NV ? T(scr(c + a * NV + ii)) : T(0);
    } else {
        T B[3][NV], dist, mu;
        contact_basis_rebuild<T>(scr, ci, B, dist, mu);
        KS_UNROLL
        for (int a = 0; a < 3; a++) b[a] = pick(B[a], i);
    }
}

// Scalar (non-contact) rows kept in registers: 3 tendon equalities + up to 6 joint limits
template <typename T> struct ScalarRows {
    T eq_aref[3], eq_R[3];
    T lim_sign[6], lim_aref[6], lim_R[6];   // sign 0 = inactive; joints: slides 0-2, proximal hinges 3,5,7
};

template <typename T, typename S, int SUBS>
KS_HD void make_constraints(const Model<T>& m, const T* qpos, const T* qvel, S scr, Team<SUBS> team, int ncon, ScalarRows<T>& r) {
    KS_UNROLL
    for (int t = 0; t < 3; t++) {
        const T c0 = m.tendon_coef[t][0], c1 = m.tendon_coef[t][1];
        T pos = c0 * qpos[3 + 2 * t] + c1 * qpos[4 + 2 * t];
        T vel = c0 * qvel[3 + 2 * t] + c1 * qvel[4 + 2 * t];
        T imp = impedance(m.solimp, pos);
        r.eq_aref[t] = -m.solref_b * vel - m.solref_k * imp * pos;
        T R = (1 - imp) / imp * m.tendon_invw[t];
        r.eq_R[t] = R > T(1e-15) ? R : T(1e-15);
    }
    KS_UNROLL
    for (int j = 0; j < 6; j++) {
        const int dof = j < 3 ? j : 3 + 2 * (j - 3);
        T lo, hi;
        bool limited = true;
        if (j < 3) { lo = m.slide_range[j][0]; hi = m.slide_range[j][1]; }
        else { lo = m.hinge_range[2 * (j - 3)][0]; hi = m.hinge_range[2 * (j - 3)][1]; limited = m.hinge_limited[2 * (j - 3)] != 0; }
        T q = qpos[dof], sign = 0, pos = 0;
        if (limited && q - lo < 0) { sign = 1; pos = q - lo; }
        if (limited && hi - q < 0) { sign = -1; pos = hi - q; }
        T imp = impedance(m.solimp, pos);
        r.lim_sign[j] = sign;
        r.lim_aref[j] = -m.solref_b * sign * qvel[dof] - m.solref_k * imp * pos;
        T R = (1 - imp) / imp * m.dof_invw[dof];
        r.lim_R[j] = R > T(1e-15) ? R : T(1e-15);
    }
    for (int ci = team.sub; ci < ncon; ci += SUBS) {
        const int o = SCR_CON + ci * CON_STRIDE;
        T B[3][NV], dist, mu;
        contact_basis<T>(scr, ci, B, dist, mu);
        if (ci < NBCACHE) {
            KS_UNROLL
            for (int a = 0; a < 3; a++) {
                KS_UNROLL
                for (int jj = 0; jj < NV; jj++) scr(SCR_BCACHE + ci * BC_STRIDE + a * NV + jj) = B[a][jj];
            }
        }
        T vb[3];
        KS_UNROLL
        for (int a = 0; a < 3; a++) {
            T v = 0;
            KS_UNROLL
            for (int j = 0; j < NV; j++) v += B[a][j] * qvel[j];
            vb[a] = v;
        }
        const int bb = (int)scr(o + 8);
        const T margin = m.pair_margin[bb >> 8];   // explicit <pair>s: the pair's own (0 in the reference's XMLs); dynamic pairs: the geoms' 0.001
        T rr = dist - margin;
        T imp = impedance(m.solimp, rr);
        // the object's inverse weight follows its per-env mass
        // (the object geoms' share follows a per-env object mass)
        const T wobj = m.pair_invw[bb >> 8][1] * (m.mass[NBODY - 1] + m.armature[9]) / (T(scr(SCR_ENVP)) + m.armature[9]);
        T w = m.pair_invw[bb >> 8][0] + wobj;
        T diag = (w + mu * mu * w) * 2 * mu * mu / m.impratio;
        T R = (1 - imp) / imp * diag;
        // rows with dist >= margin are inactive: flag with R < 0
        scr(o + 9) = (dist < margin) ? (R > T(1e-15) ? R : T(1e-15)) : T(-1);
        T base = -m.solref_k * imp * rr;
        scr(o + 10) = -m.solref_b * (vb[0] + mu * vb[1]) + base;
        scr(o + 11) = -m.solref_b * (vb[0] - mu * vb[1]) + base;
        scr(o + 12) = -m.solref_b * (vb[0] + mu * vb[2]) + base;
        scr(o + 13) = -m.solref_b * (vb[0] - mu * vb[2]) + base;
    }
    team.sync();
}

// ---------------------------------------------------------------- S6 Newton solver
// line-search step tolerance (relative): fp32 stops at 1e-5 - the piecewise-linear derivative is evaluated with
// ~1e-6 relative noise near its root, chasing machine precision only burns iterations; the Newton exit test
// bounds what an inexact step can cost.  fp64 keeps the oracle's setting.
template <typename T> KS_HD constexpr T LS_RTOL() { return sizeof(T) == 4 ? T(1e-5) : T(4.8e-16); }

// Compile-time loop: f(std::integral_constant<int, 0>) ... f(<N-1>); the DPP broadcasts need constant lanes.
template <typename F, int... Is> KS_HD void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F> KS_HD void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ---- row-distributed dense kernels (row i in team lane i % SUBS, slot i / SUBS)
// In-place Cholesky A = L L^T, right-looking: afterwards register k of a row owner holds L[row][k] (k <= row),
// rd = 1 / L[row][row].  Pivots are clamped like the oracle's.
template <typename T, int SUBS, int RPL>
KS_HD void rows_cholesky(Team<SUBS> team, const int (&row)[RPL], T (&H)[RPL][NV], T (&rd)[RPL]) {
    KS_UNROLL
    for (int rr = 0; rr < RPL; rr++) rd[rr] = 0;
    static_for<NV>([&](auto kc) {
        constexpr int k = decltype(kc)::value, ko = k % SUBS, ks = k / SUBS;
        T piv = team.template bcast<ko>(H[ks][k]);
        piv = piv > T(1e-15) ? piv : T(1e-15);
        const T rp = krsqrt(piv);
        KS_UNROLL
        for (int rr = 0; rr < RPL; rr++) {
            H[rr][k] *= rp;
            if (row[rr] == k) rd[rr] = rp;
        }
        static_for<NV - 1 - k>([&](auto jc) {
            constexpr int j = k + 1 + decltype(jc)::value, jo = j % SUBS, js = j / SUBS;
            const T Ljk = team.template bcast<jo>(H[js][k]);
            KS_UNROLL
            for (int rr = 0; rr < RPL; rr++) H[rr][j] -= H[rr][k] * Ljk;
        });
    });
}
// x = (L L^T)^-1 b with b, x distributed by row
template <typename T, int SUBS, int RPL>
KS_HD void rows_solve(Team<SUBS> team, const int (&row)[RPL], const T (&L)[RPL][NV], const T (&rd)[RPL], const T (&b)[RPL], T (&x)[RPL]) {
    T bb[RPL], y[RPL];
    KS_UNROLL
    for (int rr = 0; rr < RPL; rr++) { bb[rr] = b[rr]; y[rr] = 0; x[rr] = 0; }
    static_for<NV>([&](auto kc) {
        constexpr int k = decltype(kc)::value, ko = k % SUBS, ks = k / SUBS;
        const T yk = team.template bcast<ko>(bb[ks] * rd[ks]);
        KS_UNROLL
        for (int rr = 0; rr < RPL; rr++) {
            if (row[rr] == k) y[rr] = yk;
            bb[rr] -= L[rr][k] * yk;
        }
    });
    static_for<NV>([&](auto kc) {
        constexpr int i0 = NV - 1 - decltype(kc)::value;
        T part = 0;
        KS_UNROLL
        for (int rr = 0; rr < RPL; rr++) part += (row[rr] > i0 && row[rr] < NV) ? L[rr][i0] * x[rr] : T(0);
        const T sacc = team.sum(part);
        KS_UNROLL
        for (int rr = 0; rr < RPL; rr++)
            if (row[rr] == i0) x[rr] = (y[rr] - sacc) * rd[rr];
    });
}
// every lane gets the whole vector
template <typename T, int SUBS, int RPL>
KS_HD void rows_replicate(Team<SUBS> team, const T (&x)[RPL], T (&out)[NV]) {
    static_for<NV>([&](auto kc) {
        constexpr int k = decltype(kc)::value, ko = k % SUBS, ks = k / SUBS;
        out[k] = team.template bcast<ko>(x[ks]);
    });
}

// Newton's method on the primal problem  min_a 1/2 (a-a0)^T M (a-a0) + sum_rows s(J a - aref),  ROW-DISTRIBUTED over
// the team: row i of the 15x15 Hessian H = M + J^T D J (and entry i of the gradient, of M a, of the Cholesky
// factor, of the search direction) lives in team lane i % SUBS, slot i / SUBS.  With SUBS = 16 every lane holds
// ONE row (15 registers) instead of a private copy of the whole matrix; columns of the factor travel with DPP
// row broadcasts, dot products with DPP row sums.  With SUBS = 1 (host lane check, oracle-equivalent) a lane
// owns all rows and the same code is the serial algorithm.  Contacts are owned by lane ci % SUBS: the owner
// evaluates the pyramid rows (J.a, active set, forces) and publishes 5 numbers per contact in LDS; every lane
// then adds the contact's contribution to its own row from the cached basis Jacobian.
// Vectors a, p (and the small inputs) are replicated in every lane.
// The stage also builds the constraint rows (S5), solves M qacc_smooth = qfrc_smooth and finishes with the
// semi-implicit Euler update (S7): everything that needs the mass matrix by rows lives in one function.
template <typename T, typename S, int SUBS>
KS_FN_SOLVER void constrained_step(const Model<T>& m, T* qpos, T* qvel, T* warm, S scr, Team<SUBS> team, int ncon, int iterations, int& status,
                            float* prof = nullptr) {
    static_assert(SUBS == 1 || SUBS == 16, "row distribution: one lane or one DPP row per env");
    constexpr int RPL = (NV + SUBS - 1) / SUBS;         // rows per lane
    constexpr int CPL = (NCON_MAX + SUBS - 1) / SUBS;   // contacts per lane
    KS_T0
    ScalarRows<T> r;
    make_constraints(m, qpos, qvel, scr, team, ncon, r);
    KS_TICK(2)
    T tc0[3], tc1[3], eqD[3], limD[6];
    KS_UNROLL
    for (int t = 0; t < 3; t++) { tc0[t] = m.tendon_coef[t][0]; tc1[t] = m.tendon_coef[t][1]; eqD[t] = T(1) / r.eq_R[t]; }
    KS_UNROLL
    for (int j = 0; j < 6; j++) limD[j] = T(1) / r.lim_R[j];
    const T lead = team.sub == 0 ? T(1) : T(0);
    // own rows of M (block diagonal: hand 9x9, object 6x6) and own entries of the smooth force
    T Mrow[RPL][NV], qs[RPL];
    int row[RPL];
    KS_UNROLL
    for (int rr = 0; rr < RPL; rr++) {
        row[rr] = team.sub + rr * SUBS;
        load_dynamics_row<T>(scr, row[rr], Mrow[rr], qs[rr]);
    }
    // qacc_smooth = M^-1 qfrc_smooth
    T a[NV], a0[NV];
    {
        T L[RPL][NV], rd[RPL], x[RPL];
        KS_UNROLL
        for (int rr = 0; rr < RPL; rr++) {
            KS_UNROLL
            for (int j = 0; j < NV; j++) L[rr][j] = Mrow[rr][j];
        }
        rows_cholesky(team, row, L, rd);
        rows_solve(team, row, L, rd, qs, x);
        rows_replicate(team, x, a0);
    }
    KS_TICK(3)

    // Warm start: the cheaper of the previous substep's solution and the unconstrained one.  Both costs in ONE pass (Gauss part
    // by rows, scalar rows on the lead lane, contacts by their owners): the rows of M and each contact's 3 x 15 basis are read
    // once for the two candidates, not once each.
    {
        T x[2][NV], c[2] = {0, 0};
        KS_UNROLL
        for (int j = 0; j < NV; j++) { x[0][j] = warm[j]; x[1][j] = a0[j]; }
        KS_UNROLL
        for (int rr = 0; rr < RPL; rr++) {
            T md = 0;                                   // candidate 1 is a0 itself: its Gauss term is zero
            KS_UNROLL
            for (int j = 0; j < NV; j++) md += Mrow[rr][j] * (x[0][j] - a0[j]);
            c[0] += T(0.5) * (pick(x[0], row[rr]) - pick(a0, row[rr])) * md;
        }
        KS_UNROLL
        for (int v = 0; v < 2; v++) {
            T cs = 0;
            KS_UNROLL
            for (int t = 0; t < 3; t++) {
                T xx = tc0[t] * x[v][3 + 2 * t] + tc1[t] * x[v][4 + 2 * t] - r.eq_aref[t];
                cs += T(0.5) * xx * xx * eqD[t];
            }
            KS_UNROLL
            for (int j = 0; j < 6; j++) {
                const int dof = j < 3 ? j : 3 + 2 * (j - 3);
                T xx = r.lim_sign[j] * x[v][dof] - r.lim_aref[j];
                if (r.lim_sign[j] != 0 && xx < 0) cs += T(0.5) * xx * xx * limD[j];
            }
            c[v] += lead * cs;
        }
        KS_UNROLL
        for (int q = 0; q < CPL; q++) {
            const int ci = team.sub + q * SUBS;
            if (ci < ncon) {
                const int o = SCR_CON + ci * CON_STRIDE;
                const T R = scr(o + 9);
                if (R >= 0) {
                    T B[3][NV], dist, mu;
                    contact_basis_cached<T>(scr, ci, B, dist, mu);
                    const T D = T(1) / R;
                    const T ar[4] = {scr(o + 10), scr(o + 11), scr(o + 12), scr(o + 13)};
                    KS_UNROLL
                    for (int v = 0; v < 2; v++) {
                        T xb3[3];
                        KS_UNROLL
                        for (int bb = 0; bb < 3; bb++) {
                            T s_ = 0;
                            KS_UNROLL
                            for (int j = 0; j < NV; j++) s_ += B[bb][j] * x[v][j];
                            xb3[bb] = s_;
                        }
                        KS_UNROLL
                        for (int kk = 0; kk < 4; kk++) {
                            T xx = xb3[0] + ((kk & 1) ? -mu : mu) * xb3[1 + (kk >> 1)] - ar[kk];
                            if (xx < 0) c[v] += T(0.5) * xx * xx * D;
                        }
                    }
                }
            }
        }
        const T cw = team.sum(c[0]), cs = team.sum(c[1]);
        const bool use_warm = cw < cs;
        KS_UNROLL
        for (int j = 0; j < NV; j++) a[j] = use_warm ? x[0][j] : a0[j];
    }
    KS_TICK(13)
    T xb[CPL][3], pb[CPL][3];
    bool converged = false;
    for (int it = 0; it < iterations; it++) {
#ifdef KS_STAMP
        if (prof) prof[21] += 1.f;
#endif
        // --- contact owners: pyramid rows at a, published as (gn, gt1, gt2, active mask, D) in slots 14..18
        KS_UNROLL
        for (int q = 0; q < CPL; q++) {
            const int ci = team.sub + q * SUBS;
            if (ci < ncon) {
                const int o = SCR_CON + ci * CON_STRIDE;
                const T R = scr(o + 9);
                T gn = 0, gt1 = 0, gt2 = 0, D = 0;
                int mask = 0;
                if (R >= 0) {
                    T B[3][NV], dist, mu;
                    contact_basis_cached<T>(scr, ci, B, dist, mu);
                    KS_UNROLL
                    for (int b = 0; b < 3; b++) {
                        T v = 0;
                        KS_UNROLL
                        for (int j = 0; j < NV; j++) v += B[b][j] * a[j];
                        xb[q][b] = v;
                    }
                    D = T(1) / R;
                    T y[4];
                    KS_UNROLL
                    for (int kk = 0; kk < 4; kk++) {
                        T xx = xb[q][0] + ((kk & 1) ? -mu : mu) * xb[q][1 + (kk >> 1)] - scr(o + 10 + kk);
                        const bool act = xx < 0;
                        mask |= act ? (1 << kk) : 0;
                        y[kk] = act ? D * xx : T(0);
                    }
                    gn = y[0] + y[1] + y[2] + y[3]; gt1 = mu * (y[0] - y[1]); gt2 = mu * (y[2] - y[3]);
                }
                scr(o + 14) = gn; scr(o + 15) = gt1; scr(o + 16) = gt2; scr(o + 17) = T(mask); scr(o + 18) = D;
            }
        }
        team.sync();
        // --- own rows: gradient and Hessian
        T H[RPL][NV], g[RPL], Ma[RPL];
        KS_UNROLL
        for (int rr = 0; rr < RPL; rr++) {
            const int i = row[rr];
            T v = 0;
            KS_UNROLL
            for (int j = 0; j < NV; j++) { v += Mrow[rr][j] * a[j]; H[rr][j] = Mrow[rr][j]; }
            Ma[rr] = v - qs[rr];
            g[rr] = Ma[rr];
            KS_UNROLL
            for (int t = 0; t < 3; t++) {
                const int ip = 3 + 2 * t, id = 4 + 2 * t;
                const T xx = tc0[t] * a[ip] + tc1[t] * a[id] - r.eq_aref[t];
                const T ci_ = (i == ip) ? tc0[t] : ((i == id) ? tc1[t] : T(0));   // J[t][i]
                g[rr] += ci_ * eqD[t] * xx;
                H[rr][ip] += ci_ * eqD[t] * tc0[t];
                H[rr][id] += ci_ * eqD[t] * tc1[t];
            }
            KS_UNROLL
            for (int j = 0; j < 6; j++) {
                const int dof = j < 3 ? j : 3 + 2 * (j - 3);
                const T xx = r.lim_sign[j] * a[dof] - r.lim_aref[j];
                if (i == dof && r.lim_sign[j] != 0 && xx < 0) {
                    g[rr] += r.lim_sign[j] * limD[j] * xx;
                    H[rr][dof] += limD[j];
                }
            }
        }
        for (int ci = 0; ci < ncon; ci++) {
            const int o = SCR_CON + ci * CON_STRIDE;
            const int mask = (int)scr(o + 17);
            if (mask == 0) continue;
            const T gn = scr(o + 14), gt1 = scr(o + 15), gt2 = scr(o + 16), D = scr(o + 18);
            T B[3][NV], dist, mu;
            contact_basis_cached<T>(scr, ci, B, dist, mu);
            const T a0_ = (mask & 1) ? T(1) : T(0), a1_ = (mask & 2) ? T(1) : T(0), a2_ = (mask & 4) ? T(1) : T(0), a3_ = (mask & 8) ? T(1) : T(0);
            const T Cnn = D * (a0_ + a1_ + a2_ + a3_);
            const T Cn1 = D * mu * (a0_ - a1_), Cn2 = D * mu * (a2_ - a3_);
            const T C11 = D * mu * mu * (a0_ + a1_), C22 = D * mu * mu * (a2_ + a3_);
            KS_UNROLL
            for (int rr = 0; rr < RPL; rr++) {
                const int i = row[rr];
                T bc[3];
                contact_basis_column<T>(scr, ci, i, bc);
                const T b0 = bc[0], b1 = bc[1], b2 = bc[2];
                g[rr] += b0 * gn + b1 * gt1 + b2 * gt2;
                const T u0 = Cnn * b0 + Cn1 * b1 + Cn2 * b2, u1 = Cn1 * b0 + C11 * b1, u2 = Cn2 * b0 + C22 * b2;
                KS_UNROLL
                for (int j = 0; j < NV; j++) H[rr][j] += u0 * B[0][j] + u1 * B[1][j] + u2 * B[2][j];
            }
        }
        KS_TICK(14)
        // --- Newton direction: H p = -g
        T rd[RPL], ng[RPL], x[RPL], p[NV];
        KS_UNROLL
        for (int rr = 0; rr < RPL; rr++) ng[rr] = -g[rr];
        rows_cholesky(team, row, H, rd);
        KS_TICK(15)
        rows_solve(team, row, H, rd, ng, x);
        rows_replicate(team, x, p);
        KS_TICK(16)
        // --- line search data
        T pMa = 0, pMp = 0;
        KS_UNROLL
        for (int rr = 0; rr < RPL; rr++) {
            T mp = 0;
            KS_UNROLL
            for (int j = 0; j < NV; j++) mp += Mrow[rr][j] * p[j];
            pMa += x[rr] * Ma[rr];
            pMp += x[rr] * mp;
        }
        pMa = team.sum(pMa);
        pMp = team.sum(pMp);
        T eq_x[3], eq_p[3], lim_x[6], lim_p[6];
        KS_UNROLL
        for (int t = 0; t < 3; t++) {
            eq_x[t] = tc0[t] * a[3 + 2 * t] + tc1[t] * a[4 + 2 * t] - r.eq_aref[t];
            eq_p[t] = tc0[t] * p[3 + 2 * t] + tc1[t] * p[4 + 2 * t];
        }
        KS_UNROLL
        for (int j = 0; j < 6; j++) {
            const int dof = j < 3 ? j : 3 + 2 * (j - 3);
            lim_x[j] = r.lim_sign[j] * a[dof] - r.lim_aref[j];
            lim_p[j] = r.lim_sign[j] * p[dof];
        }
        // SUBS = 16: the nine scalar rows are dealt to the lanes (lane 0-2 one tendon equality each, lane 3-8 one joint limit
        // each, lane 9 the Gauss term) instead of all being evaluated by every lane and counted on the lead lane: a
        // line-search iteration is then one scalar row + the owned contact rows per lane.  sx + alpha * sp is the row,
        // sw its weight, s_eq: always active (equality), s_on: a limit row that exists.
        T sx = 0, sp = 0, sw = 0;
        bool s_eq = false, s_on = false;
        if constexpr (SUBS == 16) {
            KS_UNROLL
            for (int t = 0; t < 3; t++)
                if (team.sub == t) { sx = eq_x[t]; sp = eq_p[t]; sw = eqD[t]; s_eq = true; }
            KS_UNROLL
            for (int j = 0; j < 6; j++)
                if (team.sub == 3 + j) { sx = lim_x[j]; sp = lim_p[j]; sw = limD[j]; s_on = r.lim_sign[j] != 0; }
        }
        const T gauss = (SUBS == 16 && team.sub == 9) ? T(1) : T(0);
        // rows of the owned contacts along the search line: x(alpha) = rx + alpha * rj, weight rD (0 = not a row)
        T rx[CPL][4], rj[CPL][4], rD[CPL];
        KS_UNROLL
        for (int q = 0; q < CPL; q++) {
            const int ci = team.sub + q * SUBS;
            rD[q] = 0;
            KS_UNROLL
            for (int kk = 0; kk < 4; kk++) { rx[q][kk] = 0; rj[q][kk] = 0; }
            if (ci < ncon && scr(SCR_CON + ci * CON_STRIDE + 9) >= 0) {
                const int o = SCR_CON + ci * CON_STRIDE;
                T B[3][NV], dist, mu;
                contact_basis_cached<T>(scr, ci, B, dist, mu);
                KS_UNROLL
                for (int b = 0; b < 3; b++) {
                    T v = 0;
                    KS_UNROLL
                    for (int j = 0; j < NV; j++) v += B[b][j] * p[j];
                    pb[q][b] = v;
                }
                rD[q] = scr(o + 18);
                KS_UNROLL
                for (int kk = 0; kk < 4; kk++) {
                    const T sm = (kk & 1) ? -mu : mu;
                    rj[q][kk] = pb[q][0] + sm * pb[q][1 + (kk >> 1)];
                    rx[q][kk] = xb[q][0] + sm * xb[q][1 + (kk >> 1)] - scr(o + 10 + kk);
                }
            }
        }
        KS_TICK(17)
        // --- exact line search on phi'(alpha) (piecewise linear, increasing)
        T alpha = 0, lo = 0, hi = -1;
        for (int ls = 0; ls < 30; ls++) {
#if defined(KS_STAMP) && !defined(KS_STAMP_HULL)
            if (prof) prof[22] += 1.f;
#endif
            T d1, d2, mag;
            if constexpr (SUBS == 16) {
                d1 = gauss * (pMa + alpha * pMp); d2 = gauss * pMp; mag = gauss * (kabs(pMa) + kabs(alpha * pMp));
                const T xx = sx + alpha * sp, w = sw * sp;
                if (s_eq || (s_on && xx < 0)) { d1 += w * xx; d2 += w * sp; mag += kabs(w * xx); }
            } else {
                d1 = pMa + alpha * pMp; d2 = pMp; mag = kabs(pMa) + kabs(alpha * pMp);
                KS_UNROLL
                for (int t = 0; t < 3; t++) {
                    const T xx = eq_x[t] + alpha * eq_p[t], w = eqD[t] * eq_p[t];
                    d1 += w * xx; d2 += w * eq_p[t]; mag += kabs(w * xx);
                }
                KS_UNROLL
                for (int j = 0; j < 6; j++) {
                    const T xx = lim_x[j] + alpha * lim_p[j], w = limD[j] * lim_p[j];
                    if (r.lim_sign[j] != 0 && xx < 0) { d1 += w * xx; d2 += w * lim_p[j]; mag += kabs(w * xx); }
                }
                d1 *= lead; d2 *= lead; mag *= lead;
            }
            KS_UNROLL
            for (int q = 0; q < CPL; q++) {
                KS_UNROLL
                for (int kk = 0; kk < 4; kk++) {
                    const T xx = rx[q][kk] + alpha * rj[q][kk], w = rD[q] * rj[q][kk];
                    if (xx < 0) { d1 += w * xx; d2 += w * rj[q][kk]; mag += kabs(w * xx); }
                }
            }
            d1 = team.sum(d1);
            d2 = team.sum(d2);
            T magsum = 0;
            if constexpr (sizeof(T) == 4) magsum = team.sum(mag);     // (here, so that the three butterflies are issued interleaved)
            if (d2 < T(1e-15)) break;
            if (d1 < 0) lo = alpha; else hi = alpha;
            T next = alpha - d1 / d2;
            if (hi >= 0 && (next < lo || next > hi)) next = T(0.5) * (lo + hi);
            if (next < lo) next = lo;
            bool stop = kabs(next - alpha) <= LS_RTOL<T>() * kabs(alpha) || kabs(next - alpha) <= T(1e-14) * (1 + kabs(alpha));
            // fp32: the derivative is below the rounding noise of its own terms - alpha cannot be resolved further
            if constexpr (sizeof(T) == 4) stop = stop || kabs(d1) <= T(2e-6) * magsum;
            alpha = next;
            if (stop) break;
        }
        // Did any row change sides between a and a + alpha p?  If not, the cost is one quadratic on the whole step,
        // the exact line search landed on its minimiser and that is the solution: no confirming iteration needed.
        T flips = 0;
        if constexpr (SUBS == 16) {
            if (s_on && ((sx < 0) != (sx + alpha * sp < 0))) flips += T(1);
        } else {
            KS_UNROLL
            for (int j = 0; j < 6; j++)
                if (r.lim_sign[j] != 0 && ((lim_x[j] < 0) != (lim_x[j] + alpha * lim_p[j] < 0))) flips += lead;
        }
        KS_UNROLL
        for (int q = 0; q < CPL; q++) {
            KS_UNROLL
            for (int kk = 0; kk < 4; kk++)
                if (rD[q] != 0 && ((rx[q][kk] < 0) != (rx[q][kk] + alpha * rj[q][kk] < 0))) flips += T(1);
        }
        flips = team.sum(flips);
        KS_TICK(18)
        T amax = 0, dmax = 0;
        KS_UNROLL
        for (int j = 0; j < NV; j++) {
            const T da = alpha * p[j];
            a[j] += da;
            amax = kabs(a[j]) > amax ? kabs(a[j]) : amax;
            dmax = kabs(da) > dmax ? kabs(da) : dmax;
        }
        team.sync();                                   // slots 14..18 are rewritten by the next iteration
        // converged: the step just taken is below 1e-5 of the solution scale (Newton is quadratic, the
        // next step would be far smaller); lanes that are done wait for the slowest env of the wave
        if (dmax <= T(1e-5) * (1 + amax) || flips == 0) { converged = true; break; }
    }
    // the iteration cap ended the loop before the stop rule did: the acceleration is a truncated Newton iterate (sticky flag,
    // ks_get_state; tests/studies/solver_cap.py measures what a cap of 6 costs)
    if (!converged) status |= ST_NEWTON_CAP;
    KS_TICK(19)
    // --- constraint forces at the final a: owners publish (fn, ft1, ft2) in slots 14..16 (also the parity tap)
    KS_UNROLL
    for (int q = 0; q < CPL; q++) {
        const int ci = team.sub + q * SUBS;
        if (ci < ncon) {
            const int o = SCR_CON + ci * CON_STRIDE;
            const T R = scr(o + 9);
            T fn = 0, ft1 = 0, ft2 = 0;
            if (R >= 0) {
                T B[3][NV], dist, mu, xf[3];
                contact_basis_cached<T>(scr, ci, B, dist, mu);
                KS_UNROLL
                for (int b = 0; b < 3; b++) {
                    T v = 0;
                    KS_UNROLL
                    for (int j = 0; j < NV; j++) v += B[b][j] * a[j];
                    xf[b] = v;
                }
                const T D = T(1) / R;
                T f[4];
                KS_UNROLL
                for (int kk = 0; kk < 4; kk++) {
                    T xx = xf[0] + ((kk & 1) ? -mu : mu) * xf[1 + (kk >> 1)] - scr(o + 10 + kk);
                    f[kk] = xx < 0 ? -xx * D : T(0);
                }
                fn = f[0] + f[1] + f[2] + f[3]; ft1 = mu * (f[0] - f[1]); ft2 = mu * (f[2] - f[3]);
            }
            scr(o + 14) = fn; scr(o + 15) = ft1; scr(o + 16) = ft2;
        }
    }
    team.sync();
    // qfrc_c = J^T f by rows, then replicated
    T qc[RPL];
    KS_UNROLL
    for (int rr = 0; rr < RPL; rr++) {
        const int i = row[rr];
        T v = 0;
        KS_UNROLL
        for (int t = 0; t < 3; t++) {
            const int ip = 3 + 2 * t, id = 4 + 2 * t;
            const T xx = tc0[t] * a[ip] + tc1[t] * a[id] - r.eq_aref[t];
            const T ci_ = (i == ip) ? tc0[t] : ((i == id) ? tc1[t] : T(0));
            v -= ci_ * xx * eqD[t];
        }
        KS_UNROLL
        for (int j = 0; j < 6; j++) {
            const int dof = j < 3 ? j : 3 + 2 * (j - 3);
            const T xx = r.lim_sign[j] * a[dof] - r.lim_aref[j];
            if (i == dof && r.lim_sign[j] != 0 && xx < 0) v -= r.lim_sign[j] * xx * limD[j];
        }
        qc[rr] = v;
    }
    for (int ci = 0; ci < ncon; ci++) {
        const int o = SCR_CON + ci * CON_STRIDE;
        const T fn = scr(o + 14), ft1 = scr(o + 15), ft2 = scr(o + 16);
        if (fn == 0) continue;
        KS_UNROLL
        for (int rr = 0; rr < RPL; rr++) {
            T b[3];
            contact_basis_column<T>(scr, ci, row[rr], b);
            qc[rr] += b[0] * fn + b[1] * ft1 + b[2] * ft2;
        }
    }
    KS_TICK(20)
    // --- S7 Euler with implicit joint damping: (M + h D) qacc' = qfrc_smooth + qfrc_constraint, by rows
    const T h = m.dt;
    T qa[NV];
    {
        T L[RPL][NV], rd[RPL], f[RPL], x[RPL];
        KS_UNROLL
        for (int rr = 0; rr < RPL; rr++) {
            const T hd = row[rr] < NV ? h * m.damping[row[rr] < NV ? row[rr] : 0] : T(0);
            KS_UNROLL
            for (int j = 0; j < NV; j++) L[rr][j] = Mrow[rr][j] + (j == row[rr] ? hd : T(0));
            f[rr] = qs[rr] + qc[rr];
        }
        rows_cholesky(team, row, L, rd);
        rows_solve(team, row, L, rd, f, x);
        rows_replicate(team, x, qa);
    }
    KS_UNROLL
    for (int i = 0; i < NV; i++) warm[i] = a[i];
    bool finite = true;
    T qv[NV];
    KS_UNROLL
    for (int i = 0; i < NV; i++) {
        qv[i] = qvel[i] + h * qa[i];
        qvel[i] = qv[i];
        finite = finite && (qv[i] == qv[i]) && kabs(qv[i]) < T(1e10);
    }
    KS_UNROLL
    for (int i = 0; i < 12; i++) qpos[i] += h * qv[i];
    T w[3] = {qv[12], qv[13], qv[14]}, quat[4] = {qpos[12], qpos[13], qpos[14], qpos[15]};
    const T ang = norm3(w) * h;
    if (ang > T(1e-15)) {
        normalize3(w);
        const T sa = ksin(T(0.5) * ang), dq[4] = {kcos(T(0.5) * ang), w[0] * sa, w[1] * sa, w[2] * sa};
        quatmul(quat, quat, dq);
    }
    quatnormalize(quat);
    KS_UNROLL
    for (int i = 0; i < 4; i++) qpos[12 + i] = quat[i];
    if (!finite) status |= ST_NONFINITE;
    team.sync();
    KS_TICK(5)
}

// ---------------------------------------------------------------- one mj_step (forward + Euler)
struct NoHook {
    KS_HD void operator()() const {}
};
// after_kinematics(): called once the body poses of this step are in the env's scratch block (before collision / solver)
template <typename T, typename S, int SUBS, typename Hook = NoHook>
KS_HD void mj_forward_step(const Model<T>& m, const Hulls<T>& hu, T* qpos, T* qvel, T* warm, const T* ctrl, const T* R7, S scr, Team<SUBS> team,
                           int solver_iterations, bool integrate, int& ncon_out, int& status, float* prof = nullptr, PairWarm* gjk_warm = nullptr,
                           Hook after_kinematics = Hook()) {
    KS_T0
    team.sync();                                   // the previous substep's readers of the body poses are done
    dynamics_rows(m, qpos, qvel, ctrl, R7, scr, team);
    after_kinematics();
    KS_TICK(0)
    int ncon = 0;
#if defined(KS_SCRATCH_PROBE) && defined(__HIP_DEVICE_COMPILE__)
    // EXPERIMENT (tools/experiments/scratch_probe.sh): KS_SCRATCH_PROBE (a power of two) extra private-memory words per lane stored
    // before the out-of-line `collision` and loaded after it - the pattern of the register frame that lives there ("no build
    // without those stores exists to compare" - one WITH MORE of them does, and the slope of the run time over their number is what one of them costs).
    // The array is indexed with a run-time offset (always 0) so that it stays in private memory; stores and loads are independent and issue back to back
    // like a spill sequence.
    T probe_[KS_SCRATCH_PROBE];
    KS_UNROLL
    for (int i = 0; i < KS_SCRATCH_PROBE; i++) probe_[(i + (status >> 30)) & (KS_SCRATCH_PROBE - 1)] = qvel[i % NV];
#endif
    collision(m, hu, scr, team, ncon, status, gjk_warm, prof);
#if defined(KS_SCRATCH_PROBE) && defined(__HIP_DEVICE_COMPILE__)
    {
        T acc_ = T(0);
        KS_UNROLL
        for (int i = 0; i < KS_SCRATCH_PROBE; i++) acc_ += probe_[(i + (ncon >> 30)) & (KS_SCRATCH_PROBE - 1)];
        if (acc_ == T(-12345.678)) status |= ST_CONTACT_OVERFLOW;       // (never: keeps the loads)
    }
#endif
    KS_TICK(1)
    ncon_out = ncon;
    if (!integrate) return;
    constrained_step(m, qpos, qvel, warm, scr, team, ncon, solver_iterations, status, prof);
}

}  // namespace ks
