// tests/native/ks_lanecheck.cpp -- TEST-ONLY host build of the kernel source (ks_core.h / ks_obs.h /
// ks_env.h): runs ONE lane of the gfx950 kernels' code on the CPU so that `-m "not gpu"` tests can
// check the kernel logic against the fp64 oracle without a GPU.  Not part of the product library:
// libkinova_sim.so has no CPU path and fails loudly without a GPU.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../kinovagrasping_amd/csrc/ks_env.h"
#include "../../kinovagrasping_amd/csrc/ks_model_host.h"

using namespace ks;

struct LC {
    HostModel<float> f;
    HostModel<double> d;
    // lc_set_warm(1): the lane remembers its hull pairs' last queries from one lc_substep call to the next, as a lane of the stepping kernel does
    // across substeps and launches (ks_core.h: PairWarm); default: cold queries
    PairWarm gw[NPAIR_MAX];
    bool warm = false;
};

template <typename T> static Hulls<T> host_hulls(const Model<T>& m) {
    Hulls<T> h;
    for (int s = 0; s < NMESH; s++) { h.vert[s] = m.mesh_vert[s]; h.nvert[s] = m.mesh_nvert[s]; h.nvert_pad[s] = m.mesh_nvert_pad[s]; h.adj_off[s] = m.mesh_adj_off[s]; h.adj[s] = m.mesh_adj[s]; }
    hulls_set_pairs(m, h);
    static thread_local std::vector<PairRec<T>> table;
    table.assign(NPAIR_MAX, PairRec<T>{});
    for (int pi = 0; pi < m.npair; pi++) fill_pair_rec(m, h, pi, table[pi]);
    h.pair = table.data();
    return h;
}

template <typename T> struct SnapPut {
    T* s;
    void operator()(int k, T v) const { s[k] = v; }
};

template <typename T>
static int substep_t(const Model<T>& m, double* qpos, double* qvel, double* warm, const double* ctrl, const double* hq, int iters, int* ncon,
                     double* con, PairWarm* gw = nullptr) {
    std::vector<T> scrbuf(SCR_TOTAL, T(0));
    Scratch<T> scr{scrbuf.data(), 1};
    { T mass, mu; nominal_env_params(m, mass, mu); scr(SCR_ENVP) = mass; scr(SCR_ENVP + 1) = mu; }
    LaneState<T> st;
    T c[NU], q4[4], R7[9];
    for (int i = 0; i < NQ; i++) st.qpos[i] = (T)qpos[i];
    for (int i = 0; i < NV; i++) { st.qvel[i] = (T)qvel[i]; st.warm[i] = (T)warm[i]; }
    for (int i = 0; i < NU; i++) c[i] = (T)ctrl[i];
    for (int i = 0; i < 4; i++) q4[i] = (T)hq[i];
    hand_rotation(q4, R7);
    if (std::getenv("KS_LC_EXACT_R7")) {          // experiment: the hand's rotation formed in fp64 and rounded once
        double q4d[4] = {hq[0], hq[1], hq[2], hq[3]}, R7d[9];
        hand_rotation(q4d, R7d);
        for (int i = 0; i < 9; i++) R7[i] = (T)R7d[i];
    }
    int status = 0, nc = 0;
    mj_forward_step(m, host_hulls(m), st.qpos, st.qvel, st.warm, c, R7, scr, Team<1>{0}, iters, true, nc, status, nullptr, gw);
    for (int i = 0; i < NQ; i++) qpos[i] = st.qpos[i];
    for (int i = 0; i < NV; i++) { qvel[i] = st.qvel[i]; warm[i] = st.warm[i]; }
    *ncon = nc;
    if (con)
        for (int i = 0; i < nc * CON_STRIDE; i++) con[i] = scrbuf[SCR_CON + i];
    return status;
}

// EXPERIMENT (tests/studies/divergence_table.py): what a mixed-precision step would do - kinematics, mass matrix, smooth forces and the
// whole collision stage in fp64 (the state is kept in fp64 between substeps), the constraint solver and the Euler step in fp32 on the rounded scratch.
static int g_mixed_variant = 0;      // 0: fp64 state + kinematics + collision | 1: fp64 state + kinematics only (collision fp32 on the rounded poses) | 2: fp32 kinematics, fp64 collision on them | 3 / 4: as 2 for the hull pairs / the plane pairs only
static int g_state_rounding = 0;     // bit 0: qvel rounded to fp32 after every substep, bit 1: qpos (lc_set_mixed_variant(v + 10 * bits))
static int substep_mixed(const Model<double>& md, const Model<float>& mf, double* qpos, double* qvel, double* warm, const double* ctrl, const double* hq,
                         int iters, int* ncon, double* con) {
    std::vector<double> sd(SCR_TOTAL, 0.0);
    std::vector<float> sf(SCR_TOTAL, 0.f);
    Scratch<double> scrd{sd.data(), 1};
    Scratch<float> scrf{sf.data(), 1};
    { double mass, mu; nominal_env_params(md, mass, mu); scrd(SCR_ENVP) = mass; scrd(SCR_ENVP + 1) = mu; }
    double q4[4], R7[9], c[NU];
    for (int i = 0; i < 4; i++) q4[i] = hq[i];
    for (int i = 0; i < NU; i++) c[i] = ctrl[i];
    hand_rotation(q4, R7);
    int status = 0, nc = 0;
    if (g_mixed_variant == 0) {
        mj_forward_step(md, host_hulls(md), qpos, qvel, warm, c, R7, scrd, Team<1>{0}, iters, false, nc, status);       // integrate = false: stops after collision
        for (int i = 0; i < SCR_TOTAL; i++) sf[i] = (float)sd[i];
    } else if (g_mixed_variant == 1) {
        dynamics_rows(md, qpos, qvel, c, R7, scrd, Team<1>{0});
        for (int i = 0; i < SCR_TOTAL; i++) sf[i] = (float)sd[i];
        collision(mf, host_hulls(mf), scrf, Team<1>{0}, nc, status);
        for (int i = 0; i < SCR_TOTAL; i++) sd[i] = (double)sf[i];
    } else {
        float qpf[NQ], qvf[NV], cf[NU], R7f[9];
        for (int i = 0; i < NQ; i++) qpf[i] = (float)qpos[i];
        for (int i = 0; i < NV; i++) qvf[i] = (float)qvel[i];
        for (int i = 0; i < NU; i++) cf[i] = (float)c[i];
        for (int i = 0; i < 9; i++) R7f[i] = (float)R7[i];
        { float mass, mu; nominal_env_params(mf, mass, mu); scrf(SCR_ENVP) = mass; scrf(SCR_ENVP + 1) = mu; }
        dynamics_rows(mf, qpf, qvf, cf, R7f, scrf, Team<1>{0});
        for (int i = 0; i < SCR_TOTAL; i++) sd[i] = (double)sf[i];
        if (g_mixed_variant >= 5) {
            // 5: everything fp32 but the ACCUMULATION of the state (below) | 6: qpos accumulated in fp64, qvel rounded to fp32 after every substep | 7: the other way round
            collision(mf, host_hulls(mf), scrf, Team<1>{0}, nc, status);
            for (int i = 0; i < SCR_TOTAL; i++) sd[i] = (double)sf[i];
        } else if (g_mixed_variant == 2) {
            collision(md, host_hulls(md), scrd, Team<1>{0}, nc, status);
            for (int i = 0; i < SCR_TOTAL; i++) sf[i] = (float)sd[i];
        } else {
            // 3: hull pairs from the fp64 collision, plane pairs from the fp32 one | 4: the other way round (both on the same fp32 poses; the
            // two contact lists are in pair order, field 8 of a record names its pair)
            int ncd = 0, ncf = 0, st2 = 0;
            collision(md, host_hulls(md), scrd, Team<1>{0}, ncd, status);
            collision(mf, host_hulls(mf), scrf, Team<1>{0}, ncf, st2);
            std::vector<float> out((size_t)NCON_MAX * CON_STRIDE, 0.f);
            int id = 0, jf = 0, n = 0;
            auto pair_of = [](double w) { return (int)w / 256; };
            while ((id < ncd || jf < ncf) && n < NCON_MAX) {
                const int pd = id < ncd ? pair_of(sd[SCR_CON + id * CON_STRIDE + 8]) : 1 << 30, pf = jf < ncf ? pair_of((double)sf[SCR_CON + jf * CON_STRIDE + 8]) : 1 << 30;
                const bool take_d = pd <= pf;
                const int pi = take_d ? pd : pf;
                const bool want_d = is_plane_pair(md, pi) ? g_mixed_variant == 4 : g_mixed_variant == 3;
                if (take_d) { if (want_d) { for (int f = 0; f < CON_STRIDE; f++) out[n * CON_STRIDE + f] = (float)sd[SCR_CON + id * CON_STRIDE + f]; n++; } id++; }
                else        { if (!want_d) { for (int f = 0; f < CON_STRIDE; f++) out[n * CON_STRIDE + f] = sf[SCR_CON + jf * CON_STRIDE + f]; n++; } jf++; }
            }
            for (int i = 0; i < n * CON_STRIDE; i++) { sf[SCR_CON + i] = out[i]; sd[SCR_CON + i] = out[i]; }
            nc = n;
        }
    }
    float qp[NQ], qv[NV], qw[NV];
    for (int i = 0; i < NQ; i++) qp[i] = (float)qpos[i];
    for (int i = 0; i < NV; i++) { qv[i] = (float)qvel[i]; qw[i] = (float)warm[i]; }
    float qv0[NV];
    for (int i = 0; i < NV; i++) qv0[i] = qv[i];
    constrained_step(mf, qp, qv, qw, scrf, Team<1>{0}, nc, iters, status);
    // the state stays in fp64: the fp32 step's velocity INCREMENT is applied to the fp64 velocity, positions integrated in fp64 from it
    const double h = (double)mf.dt;
    for (int i = 0; i < NV; i++) { qvel[i] += (double)qv[i] - (double)qv0[i]; warm[i] = qw[i]; }
    for (int i = 0; i < 12; i++) qpos[i] += h * qvel[i];
    {   // free joint: quaternion advanced by the body-frame angular velocity, as ko_physics.c / constrained_step do
        const double w3[3] = {qvel[12], qvel[13], qvel[14]};
        const double ang = std::sqrt(w3[0] * w3[0] + w3[1] * w3[1] + w3[2] * w3[2]) * h;
        if (ang > 1e-15) {
            const double sa = std::sin(0.5 * ang) / (ang / h), ca = std::cos(0.5 * ang);
            const double dq[4] = {ca, w3[0] * sa, w3[1] * sa, w3[2] * sa}, a0 = qpos[12], a1 = qpos[13], a2 = qpos[14], a3 = qpos[15];
            qpos[12] = a0 * dq[0] - a1 * dq[1] - a2 * dq[2] - a3 * dq[3];
            qpos[13] = a0 * dq[1] + a1 * dq[0] + a2 * dq[3] - a3 * dq[2];
            qpos[14] = a0 * dq[2] - a1 * dq[3] + a2 * dq[0] + a3 * dq[1];
            qpos[15] = a0 * dq[3] + a1 * dq[2] - a2 * dq[1] + a3 * dq[0];
        }
        const double nq = std::sqrt(qpos[12] * qpos[12] + qpos[13] * qpos[13] + qpos[14] * qpos[14] + qpos[15] * qpos[15]);
        for (int i = 12; i < 16; i++) qpos[i] /= nq;
    }
    if (g_mixed_variant == 6 || (g_state_rounding & 1)) for (int i = 0; i < NV; i++) qvel[i] = (double)(float)qvel[i];
    if (g_mixed_variant == 7 || (g_state_rounding & 2)) for (int i = 0; i < NQ; i++) qpos[i] = (double)(float)qpos[i];
    *ncon = nc;
    if (con)
        for (int i = 0; i < nc * CON_STRIDE; i++) con[i] = sd[SCR_CON + i];
    return status;
}

template <typename T>
static int env_step_t(const Model<T>& m, double* qpos, double* qvel, double* warm, const double* hq, const double* act, int frame_skip, int iters,
                      double* obs, double* reward, int* done, double* rays_out, int do_reset) {
    std::vector<T> scrbuf(SCR_TOTAL, T(0)), snap(SNAP_TOTAL, T(0));
    Scratch<T> scr{scrbuf.data(), 1};
    { T mass, mu; nominal_env_params(m, mass, mu); scr(SCR_ENVP) = mass; scr(SCR_ENVP + 1) = mu; }
    LaneState<T> st;
    T q4[4], a4[4];
    for (int i = 0; i < NQ; i++) st.qpos[i] = (T)qpos[i];
    for (int i = 0; i < NV; i++) { st.qvel[i] = (T)qvel[i]; st.warm[i] = (T)warm[i]; }
    for (int i = 0; i < 4; i++) { q4[i] = (T)hq[i]; a4[i] = act ? (T)act[i] : T(0); }
    int status = 0, nc = 0;
    SnapPut<T> put{snap.data()};
    if (do_reset) {
        T q0[NQ];
        for (int i = 0; i < NQ; i++) q0[i] = (T)qpos[i];
        lane_reset(m, st, q4, q0, scr, put);
    } else {
        // as on the GPU: GJK warm-started from the previous substep's simplex within the env-step
        PairWarm gw[NPAIR_MAX];
        std::memset(gw, 0, sizeof gw);
        lane_env_step(m, host_hulls(m), st, q4, a4, scr, Team<1>{0}, put, frame_skip, iters, nc, status, nullptr, (T*)nullptr, gw);
    }
    Col<T> sc{snap.data(), 1};
    T rays[NRAY];
    for (int i = 0; i < NRAY; i++) rays[i] = rangefinder(m, sc, i);
    T o[NOBS], rew, info[3];
    bool lifted;
    build_obs(m, sc, rays, [&](int j, T v) { o[j] = v; }, rew, lifted, info);
    for (int i = 0; i < NQ; i++) qpos[i] = st.qpos[i];
    for (int i = 0; i < NV; i++) { qvel[i] = st.qvel[i]; warm[i] = st.warm[i]; }
    for (int i = 0; i < NOBS; i++) obs[i] = o[i];
    for (int i = 0; i < NRAY; i++) rays_out[i] = rays[i];
    *reward = rew;
    *done = lifted ? 1 : 0;
    return status;
}

#ifdef KS_PLANE_HOOK
// experiment: the deepest vertex of a plane pair from the fp64 tables on the fp32 lane's pose (the oracle's rule: lowest index within 1e-12)
static const Model<double>* g_hook_model = nullptr;
static int plane_hook(int g2, const double* R, const double* p, int coarse) {
    const Model<double>& m = *g_hook_model;
    const int mesh = m.geom_mesh[g2], nv = m.mesh_nvert[mesh];
    const double* V = m.mesh_vert[mesh];
    const double ln[3] = {R[6], R[7], R[8]}, cdist = p[2];
    double bd = 1e300;
    for (int i = 0; i < nv; i++) { const double d = cdist + V[4 * i] * ln[0] + V[4 * i + 1] * ln[1] + V[4 * i + 2] * ln[2]; if (d < bd) bd = d; }
    for (int i = 0; i < nv; i++) { const double d = cdist + V[4 * i] * ln[0] + V[4 * i + 1] * ln[1] + V[4 * i + 2] * ln[2]; if (d <= bd + 1e-12) return i; }
    return coarse;
}
#endif
#ifdef KS_PATH_STUDY
// tests/studies/path_stability.py: the sequence of support pairs of every penetration query, compared with the pair's previous query
namespace ks { void (*ks_path_hook)(const void*, int, int, int) = nullptr; const void* ks_path_pair = nullptr; }
#include <map>
struct PathRec { std::vector<int> cur, prev; long prev_tick = -1; };
static std::map<const void*, PathRec> g_paths;
static long g_tick = 0;
static std::vector<int> g_path_log;     // per query: tick gap to the pair's previous query, its length, this length, common prefix, same result
static void path_hook(const void* pair, int state, int i1, int i2) {
    PathRec& r = g_paths[pair];
    if (state >= 0) { r.cur.push_back(state | (i1 << 3) | (i2 << 17)); return; }
    r.cur.push_back(state);
    int k = 0;
    while (k < (int)r.cur.size() && k < (int)r.prev.size() && r.cur[k] == r.prev[k]) k++;
    g_path_log.push_back(r.prev_tick < 0 ? -1 : (int)(g_tick - r.prev_tick)); g_path_log.push_back((int)r.prev.size()); g_path_log.push_back((int)r.cur.size());
    g_path_log.push_back(k); g_path_log.push_back((int)(g_tick));
    r.prev.swap(r.cur); r.cur.clear(); r.prev_tick = g_tick;
}
extern "C" {
void lc_path_reset() { g_paths.clear(); g_path_log.clear(); g_tick = 0; ks::ks_path_hook = path_hook; }
void lc_path_tick() { g_tick++; }
int lc_path_log(int* out, int cap) { int n = (int)g_path_log.size(); for (int i = 0; i < n && i < cap; i++) out[i] = g_path_log[i]; return n; }
}
#endif
extern "C" {
void* lc_create(const void* blob, size_t n) {
    LC* h = new LC();
    if (!parse_model<float>(blob, n, h->f) || !parse_model<double>(blob, n, h->d)) {
        std::fprintf(stderr, "lc_create: %s %s\n", h->f.error.c_str(), h->d.error.c_str());
        delete h;
        return nullptr;
    }
    if (std::getenv("KS_LC_ROUND_TABLES"))        // experiment: the fp64 model's hull vertices rounded to fp32 (the tables the fp32 product holds)
        for (int s = 0; s < NMESH; s++)
            for (auto& v : h->d.vert[s]) v = (double)(float)v;
    return h;
}
void lc_destroy(void* h) { delete (LC*)h; }
int lc_substep(void* h, int prec, double* qpos, double* qvel, double* warm, const double* ctrl, const double* hq, int iters, int* ncon, double* con) {
    LC* l = (LC*)h;
#ifdef KS_PLANE_HOOK
    g_hook_model = &l->d.m; ks::ks_plane_hook = prec == 64 ? nullptr : plane_hook;
#endif
    if (prec == 6432) return substep_mixed(l->d.m, l->f.m, qpos, qvel, warm, ctrl, hq, iters, ncon, con);
    return prec == 64 ? substep_t<double>(l->d.m, qpos, qvel, warm, ctrl, hq, iters, ncon, con)
                      : substep_t<float>(l->f.m, qpos, qvel, warm, ctrl, hq, iters, ncon, con, l->warm ? l->gw : nullptr);
}
int lc_env_step(void* h, int prec, double* qpos, double* qvel, double* warm, const double* hq, const double* act, int frame_skip, int iters,
                double* obs, double* reward, int* done, double* rays) {
    LC* l = (LC*)h;
    return prec == 64 ? env_step_t<double>(l->d.m, qpos, qvel, warm, hq, act, frame_skip, iters, obs, reward, done, rays, 0)
                      : env_step_t<float>(l->f.m, qpos, qvel, warm, hq, act, frame_skip, iters, obs, reward, done, rays, 0);
}
int lc_reset_obs(void* h, int prec, double* qpos, double* qvel, double* warm, const double* hq, double* obs, double* reward, int* done, double* rays) {
    LC* l = (LC*)h;
    return prec == 64 ? env_step_t<double>(l->d.m, qpos, qvel, warm, hq, nullptr, 0, 0, obs, reward, done, rays, 1)
                      : env_step_t<float>(l->f.m, qpos, qvel, warm, hq, nullptr, 0, 0, obs, reward, done, rays, 1);
}
void lc_set_warm(void* h, int on) { LC* l = (LC*)h; l->warm = on != 0; std::memset(l->gw, 0, sizeof l->gw); }
void lc_set_mixed_variant(int v) { g_mixed_variant = v % 10; g_state_rounding = v / 10; }
int lc_con_stride() { return CON_STRIDE; }
int lc_ncon_max() { return NCON_MAX; }
}
