// tests/native/ks_lanecheck.cpp -- TEST-ONLY host build of the kernel source (ks_core.h / ks_obs.h /
// ks_env.h): runs ONE lane of the gfx950 kernels' code on the CPU so that `-m "not gpu"` tests can
// check the kernel logic against the fp64 oracle without a GPU.  Not part of the product library:
// libkinova_sim.so has no CPU path and fails loudly without a GPU.
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
                     double* con) {
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
    int status = 0, nc = 0;
    mj_forward_step(m, host_hulls(m), st.qpos, st.qvel, st.warm, c, R7, scr, Team<1>{0}, iters, true, nc, status);
    for (int i = 0; i < NQ; i++) qpos[i] = st.qpos[i];
    for (int i = 0; i < NV; i++) { qvel[i] = st.qvel[i]; warm[i] = st.warm[i]; }
    *ncon = nc;
    if (con)
        for (int i = 0; i < nc * CON_STRIDE; i++) con[i] = scrbuf[SCR_CON + i];
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
extern "C" {
void* lc_create(const void* blob, size_t n) {
    LC* h = new LC();
    if (!parse_model<float>(blob, n, h->f) || !parse_model<double>(blob, n, h->d)) {
        std::fprintf(stderr, "lc_create: %s %s\n", h->f.error.c_str(), h->d.error.c_str());
        delete h;
        return nullptr;
    }
    return h;
}
void lc_destroy(void* h) { delete (LC*)h; }
int lc_substep(void* h, int prec, double* qpos, double* qvel, double* warm, const double* ctrl, const double* hq, int iters, int* ncon, double* con) {
    LC* l = (LC*)h;
#ifdef KS_PLANE_HOOK
    g_hook_model = &l->d.m; ks::ks_plane_hook = prec == 64 ? nullptr : plane_hook;
#endif
    return prec == 64 ? substep_t<double>(l->d.m, qpos, qvel, warm, ctrl, hq, iters, ncon, con)
                      : substep_t<float>(l->f.m, qpos, qvel, warm, ctrl, hq, iters, ncon, con);
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
int lc_con_stride() { return CON_STRIDE; }
int lc_ncon_max() { return NCON_MAX; }
}
