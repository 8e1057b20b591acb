// ks_obs.h -- rangefinder rays (S8), 82-d observation (O1,O2), reward / termination (O3).
//
// Replaces kinova_gripper_env.py:_get_obs (438-534) with its helpers (274-343, 356-362, 538-608)
// and _get_reward (631-687); the rangefinder part replaces MuJoCo's mj_sensorPos for the 17
// <rangefinder> sensors (XML:265-288).
//
// Inputs are the body poses that mj_forward computed at the START of the last substep (the
// reference never calls sim.forward() after its 15th sim.step(), so everything the observation
// reads lags qpos by one substep) -- the "snapshot".  KS_HD: lane-checked on the CPU too.
#pragma once
#include <string.h>

#include "ks_model.h"

namespace ks {

constexpr int SNAP_BP = 0;           // body poses b = 2..9, 12 each
constexpr int SNAP_JPOS = 96;        // 9 jointpos sensors: slides, proximal 1-3, distal 1-3
constexpr int SNAP_TOTAL = 105;

// strided view of one env's column in a [K][N] struct-of-arrays buffer
template <typename T> struct Col {
    const T* base;
    long stride;
    KS_HD T operator()(int k) const { return base[(long)k * stride]; }
};

template <typename T, typename C> KS_HD void snap_body(C snap, int b, T* R, T* p) {
    const int o = SNAP_BP + (b - 2) * 12;
    KS_UNROLL
    for (int j = 0; j < 9; j++) R[j] = snap(o + j);
    KS_UNROLL
    for (int j = 0; j < 3; j++) p[j] = snap(o + 9 + j);
}

// slab test of a ray (origin o, direction d) against an axis-aligned box: entry parameter t0 >= 0 when the
// parametric overlap [t0, t1] intersects [0, tmax], -1 when the box is missed
template <typename T> KS_HD T ray_box_entry(const T* o, const T* d, const T* lo, const T* hi, T tmax) {
    T t0 = 0, t1 = tmax;
    KS_UNROLL
    for (int a = 0; a < 3; a++) {
        if (kabs(d[a]) < T(1e-15)) {
            if (o[a] < lo[a] || o[a] > hi[a]) return T(-1);
        } else {
            T ta = (lo[a] - o[a]) / d[a], tb = (hi[a] - o[a]) / d[a];
            if (ta > tb) { T w = ta; ta = tb; tb = w; }
            t0 = ta > t0 ? ta : t0;
            t1 = tb < t1 ? tb : t1;
            if (t0 > t1) return T(-1);
        }
    }
    return t0;
}
template <typename T> KS_HD bool ray_box(const T* o, const T* d, const T* lo, const T* hi, T tmax) { return ray_box_entry(o, d, lo, hi, tmax) >= 0; }

// entry parameter of a float32 box (6 values: min xyz, max xyz), padded outwards by 1e-6: the box test must never reject
// a triangle the exhaustive oracle would hit
template <typename T> KS_HD T bvh_box_entry(const float* bx, const T* lp, const T* lv, T tmax) {
    T lo[3] = {T(bx[0]) - T(1e-6), T(bx[1]) - T(1e-6), T(bx[2]) - T(1e-6)};
    T hi[3] = {T(bx[3]) + T(1e-6), T(bx[4]) + T(1e-6), T(bx[5]) + T(1e-6)};
    return ray_box_entry(lp, lv, lo, hi, tmax);
}
// The same test for the traversal, with the reciprocals of the direction computed once per walk (inv[a] = 1 / d[a],
// par[a]: the ray is parallel to the slab): multiplications instead of 6 divisions per box, no branches.  A product with
// a rounded reciprocal is within 1.5 ulp of the quotient; the 1e-6 padding of the boxes is orders of magnitude wider.
template <typename T> KS_HD T bvh_box_entry_inv(const float* bx, const T* lp, const T* inv, const bool* par, T tmax) {
    T t0 = 0, t1 = tmax;
    bool miss = false;
    KS_UNROLL
    for (int a = 0; a < 3; a++) {
        const T lo = T(bx[a]) - T(1e-6), hi = T(bx[3 + a]) + T(1e-6);
        const T ta = (lo - lp[a]) * inv[a], tb = (hi - lp[a]) * inv[a];
        const T tn = ta < tb ? ta : tb, tf = ta < tb ? tb : ta;
        if (par[a]) miss = miss || lp[a] < lo || lp[a] > hi;
        else { t0 = tn > t0 ? tn : t0; t1 = tf < t1 ? tf : t1; }
    }
    return (miss || t0 > t1) ? T(-1) : t0;
}
// Ray vs mesh geom, MuJoCo's mj_rayMesh semantics: bounding-box pre-test (geom_size about the geom origin),
// then the faces of the ORIGINAL triangle mesh, both orientations, nearest t >= 0 (-1 = miss).  The faces
// are visited through a bounding-volume hierarchy of 4-wide nodes (one 128-byte record holds four children's boxes),
// nearer child first, so that the nearest hit found so far prunes the rest; the result is the minimum over all
// faces, exactly what the exhaustive oracle computes.
// `bound(best)`: the pruning distance for the traversal, given this lane's nearest hit so far (-1 = none).  The serial
// code passes its own hit through; k_rays takes the minimum over the lanes that cast the SAME ray at the other geoms -
// a near hit on the object prunes the walk through the 28 000-triangle palm behind it.  The minimum over geoms of the
// nearest hits is unchanged by that.
struct OwnBound {
    template <typename T> KS_HD T operator()(T best) const { return best < 0 ? Lim<T>::big : best; }
};

// ray vs one triangle (9 floats), both orientations: parameter t >= 0 of the hit, -1 = miss
template <typename T> KS_HD T ray_tri(const float* v, const T* lp, const T* lv) {
    T v0[3] = {T(v[0]), T(v[1]), T(v[2])}, e1[3] = {T(v[3]) - v0[0], T(v[4]) - v0[1], T(v[5]) - v0[2]};
    T e2[3] = {T(v[6]) - v0[0], T(v[7]) - v0[1], T(v[8]) - v0[2]}, pv[3], tv[3], qv[3];
    cross3(pv, lv, e2);
    T det = dot3(e1, pv);
    if (kabs(det) < T(1e-30)) return T(-1);
    T inv = T(1) / det;
    sub3(tv, lp, v0);
    T u = dot3(tv, pv) * inv;
    if (u < 0 || u > 1) return T(-1);
    cross3(qv, tv, e1);
    T ww = dot3(lv, qv) * inv;
    if (ww < 0 || u + ww > 1) return T(-1);
    T tt = dot3(e2, qv) * inv;
    return tt >= 0 ? tt : T(-1);
}

// Traversal stack of ray_mesh: the far child of a node waits here with its entry parameter.  RAY_STACK bounds the depth
// of the hierarchy (checked when the model is loaded).  LocalStack keeps it in the lane's own registers / stack frame
// (serial code, host); k_rays keeps it in LDS (LdsStack) so that the kernel needs half the VGPRs and twice the waves
// fit on a SIMD - the walk is a chain of dependent L2 reads, and occupancy is what hides them.
template <typename T> struct LocalStack {
    int node[RAY_STACK];
    T t[RAY_STACK];
    int sp = 0;
    KS_HD void push(int n, T tt) { if (sp < RAY_STACK) { node[sp] = n; t[sp] = tt; sp++; } }
    KS_HD bool pop(int& n, T& tt) { if (sp == 0) return false; sp--; n = node[sp]; tt = t[sp]; return true; }
};
// one 32-bit word per entry, lane-interleaved (entry k of lane l at base[k * stride]): node id in the low 16 bits
// (< 65536 nodes, checked at load), the entry parameter as the top 16 bits of its float32 pattern minus one unit -
// always below the true value, so an entry is never pruned that the exact value would keep
template <typename T> struct LdsStack {
    KS_LDS unsigned* base;
    int stride;
    int sp = 0;
    KS_HD void push(int n, T tt) {
        if (sp < RAY_STACK) {
            unsigned b = (unsigned)float_bits((float)tt);
            b = b >= 0x10000u ? b - 0x10000u : 0u;
            base[sp * stride] = (b & 0xffff0000u) | (unsigned)n;
            sp++;
        }
    }
    KS_HD bool pop(int& n, T& tt) {
        if (sp == 0) return false;
        sp--;
        const unsigned w = base[sp * stride];
        n = (int)(w & 0xffffu);
        tt = (T)bits_float((int)(w & 0xffff0000u));
        return true;
    }
};

// The walk as a resumable object: start() does the bounding-box pre-test and the per-walk constants, step() visits ONE
// node and says whether there is another.  ray_mesh below just loops; the stepping kernel's in-step ray pass (ks_api.hip:
// wg_rays) interleaves the walks of different tasks on one lane, a node at a time.
template <typename T, typename Bound = OwnBound, typename Stack = LocalStack<T>> struct RayWalk {
    const float* tri;
    const float* wnode;
    T lp[3], lv[3], inv[3];
    bool par[3];
    T best;
    int node;
    Bound bound;
    Stack stack;
#ifdef KS_RAY_COUNT
    int visits_;
#endif
    KS_HD bool start(const float* tri_, const float* wnode_, const T* size, const T* lp_, const T* lv_, Bound bound_, Stack stack_) {
        T lo[3] = {-size[0], -size[1], -size[2]}, hi[3] = {size[0], size[1], size[2]};
        if (!ray_box(lp_, lv_, lo, hi, Lim<T>::big)) return false;
        tri = tri_; wnode = wnode_; bound = bound_; stack = stack_;
        best = T(-1);
        node = 0;
        KS_UNROLL
        for (int a = 0; a < 3; a++) { lp[a] = lp_[a]; lv[a] = lv_[a]; par[a] = kabs(lv_[a]) < T(1e-15); inv[a] = par[a] ? T(0) : T(1) / lv_[a]; }
#ifdef KS_RAY_COUNT
        visits_ = 0;
#endif
        return true;
    }
    KS_HD bool step() {
#ifdef KS_RAY_COUNT
        visits_++;
#endif
        // one 4-wide node = one 128-byte line: 4 child boxes + 4 child words (ks_model_host.h)
        float w[28];
        KS_UNROLL
        for (int i = 0; i < 28; i++) w[i] = wnode[32 * (long)node + i];
        const T tmax = bound(best);
        int id[4];
        T te[4];
        KS_UNROLL
        for (int k = 0; k < 4; k++) {
            id[k] = float_bits(w[24 + k]);
            te[k] = id[k] == RAY_EMPTY ? T(-1) : bvh_box_entry_inv(w + 6 * k, lp, inv, par, tmax);
        }
        // leaves among the children first: their hits tighten the bound for the rest
        KS_UNROLL
        for (int k = 0; k < 4; k++) {
            if (te[k] >= 0 && id[k] < 0) {
                const int code = -id[k] - 1, first = code >> 3, cnt = code & 7;
                for (int i = first; i < first + cnt; i++) {
                    const T tt = ray_tri(&tri[9 * (long)i], lp, lv);
                    if (tt >= 0 && (best < 0 || tt < best)) best = tt;
                }
                te[k] = T(-1);
            }
        }
        // inner children still in front of the nearest hit: nearest first, the others wait on the stack (farthest pushed first)
        KS_UNROLL
        for (int k = 0; k < 4; k++)
            if (te[k] >= 0 && best >= 0 && te[k] > best) te[k] = T(-1);
        // sort the (<= 4) candidates by entry parameter, misses (-1 -> +big) last: 5 compare-exchanges
        T key[4];
        KS_UNROLL
        for (int k = 0; k < 4; k++) key[k] = te[k] >= 0 ? te[k] : Lim<T>::big;
#define KS_CSWAP(i, j) { const bool sw = key[j] < key[i]; const T tk = sw ? key[j] : key[i]; key[j] = sw ? key[i] : key[j]; key[i] = tk; \
                         const int ti = sw ? id[j] : id[i]; id[j] = sw ? id[i] : id[j]; id[i] = ti; }
        KS_CSWAP(0, 1) KS_CSWAP(2, 3) KS_CSWAP(0, 2) KS_CSWAP(1, 3) KS_CSWAP(1, 2)
#undef KS_CSWAP
        KS_UNROLL
        for (int k = 3; k >= 1; k--)
            if (key[k] < Lim<T>::big) stack.push(id[k], key[k]);
        if (key[0] < Lim<T>::big) { node = id[0]; return true; }
        // next stacked node whose entry is still in front of the nearest hit
        bool found = false;
        int pn;
        T pt;
        while (stack.pop(pn, pt)) {
            if (best < 0 || pt <= best) { node = pn; found = true; break; }
        }
        return found;
    }
};

template <typename T, typename Bound = OwnBound, typename Stack = LocalStack<T>>
KS_HD T ray_mesh(const float* tri, const float* wnode, const T* size, const T* lp, const T* lv, Bound bound = Bound(), Stack stack = Stack()) {
    RayWalk<T, Bound, Stack> w;
    if (!w.start(tri, wnode, size, lp, lv, bound, stack)) return T(-1);
    while (w.step()) {}
#ifdef KS_RAY_COUNT
    return T(w.visits_);
#endif
    return w.best;
}

// ---- one rangefinder = ray from site `si` along its +z against the ground plane and the 8 mesh geoms (those of the
// site's own body excluded), nearest hit.  The three pieces below are what the serial loop (rangefinder) and the
// one-geom-per-lane kernel (k_rays) share; the minimum over geoms does not depend on the order.
template <typename T, typename C> KS_HD int ray_origin(const Model<T>& m, C snap, int si, T* pnt, T* vec) {
    T Rb[9], pb[3], t[3];
    const int sb = m.site_body[si];
    snap_body<T>(snap, sb, Rb, pb);
    mulRv(t, Rb, m.site_pos[si]);
    add3(pnt, pb, t);
    mulRv(vec, Rb, m.site_z[si]);
    return sb;
}
// ground plane z = 0 with finite half-size (XML:148); -1 = miss.  Front face only (mju_rayGeom's plane case): the ray must
// point down at the +z side; from below the floor nothing is seen.
template <typename T> KS_HD T ray_ground(const Model<T>& m, const T* pnt, const T* vec) {
    if (vec[2] < -T(1e-15)) {
        T tt = -pnt[2] / vec[2];
        if (tt >= 0) {
            T x = pnt[0] + tt * vec[0], y = pnt[1] + tt * vec[1];
            if (kabs(x) <= m.geom_size[0][0] && kabs(y) <= m.geom_size[0][1]) return tt;
        }
    }
    return T(-1);
}
// the ray (origin pnt, direction vec, world frame) in the frame of geom g
template <typename T, typename C> KS_HD void ray_to_geom(const Model<T>& m, C snap, int g, const T* pnt, const T* vec, T* lp, T* lv) {
    T R[9], p[3], Rg[9], pg[3], t[3];
    snap_body<T>(snap, m.geom_body[g], R, p);
    mulRR(Rg, R, m.geom_R[g]);
    mulRv(t, R, m.geom_pos[g]);
    add3(pg, p, t);
    sub3(t, pnt, pg);
    mulRtv(lp, Rg, t);
    mulRtv(lv, Rg, vec);
}
// does the ray pass through geom g's bounding box at all (the first test of ray_mesh)
template <typename T, typename C> KS_HD bool ray_may_hit_geom(const Model<T>& m, C snap, int g, const T* pnt, const T* vec) {
    T lp[3], lv[3];
    ray_to_geom(m, snap, g, pnt, vec, lp, lv);
    const T* size = m.geom_size[g];
    T lo[3] = {-size[0], -size[1], -size[2]}, hi[3] = {size[0], size[1], size[2]};
    return ray_box(lp, lv, lo, hi, Lim<T>::big);
}
template <typename T, typename C, typename Bound = OwnBound, typename Stack = LocalStack<T>>
KS_HD T ray_geom(const Model<T>& m, C snap, int g, const T* pnt, const T* vec, Bound bound = Bound(), Stack stack = Stack()) {
    T lp[3], lv[3];
    ray_to_geom(m, snap, g, pnt, vec, lp, lv);
    const int mesh = m.geom_mesh[g];
    return ray_mesh(m.mesh_tri[mesh], m.mesh_bvh_box[mesh], m.geom_size[g], lp, lv, bound, stack);
}
// nearer of two ray results (-1 = miss)
template <typename T> KS_HD T ray_nearer(T a, T b) { return (b >= 0 && (a < 0 || b < a)) ? b : a; }

template <typename T, typename C> KS_HD T rangefinder(const Model<T>& m, C snap, int si) {
    T pnt[3], vec[3];
    const int sb = ray_origin(m, snap, si, pnt, vec);
    T best = ray_ground(m, pnt, vec);
    for (int g = 1; g < m.ngeom; g++) {
        if (m.geom_body[g] == sb) continue;
        best = ray_nearer(best, ray_geom(m, snap, g, pnt, vec));
    }
    return best;
}

template <typename T> KS_HD T tri_area(const T* a, const T* b) {
    T c[3];
    cross3(c, a, b);
    return norm3(c) * T(0.5);
}

template <typename T> KS_HD T dot_product20(const T* obj, const T* hand) {
    T ox = kabs(obj[0] - hand[0]), oy = kabs(obj[1] - hand[1]);
    T on = ksqrt(ox * ox + oy * oy);
    T cx = kabs(hand[0]), cy = kabs(hand[1]);
    T cn = ksqrt(cx * cx + cy * cy);
    T d = (ox / on) * (cx / cn) + (oy / on) * (cy / cn);
    // d^20 by repeated squaring (d in [0,1]); same value as pow(d,20) to rounding
    T d2 = d * d, d4 = d2 * d2, d8 = d4 * d4, d16 = d8 * d8;
    return d16 * d4;
}

// palm transform (ENV:274-288): T3 = (R_palm C)^T rows, wrist = p_palm + T3^T [-0.009, 0.048, 0]
template <typename T> KS_HD void palm_transform(const T* Rpalm, const T* ppalm, T* T3, T* wrist) {
    // (R C) with C = [[0,0,1],[-1,0,0],[0,-1,0]]: columns of RC = (-R[:,1], -R[:,2], R[:,0])
    KS_UNROLL
    for (int i = 0; i < 3; i++) {
        T3[0 * 3 + i] = -Rpalm[3 * i + 1];
        T3[1 * 3 + i] = -Rpalm[3 * i + 2];
        T3[2 * 3 + i] = Rpalm[3 * i + 0];
    }
    const T off[3] = {T(-0.009), T(0.048), T(0)};
    KS_UNROLL
    for (int i = 0; i < 3; i++) wrist[i] = ppalm[i] + T3[i] * off[0] + T3[3 + i] * off[1] + T3[6 + i] * off[2];
}

// action (4) -> 9 controls (ENV:1495-1535); R_palm is constant over an episode for this model
template <typename T> KS_HD void action_to_ctrl(const T* T3, const T* act4, T* ctrl) {
    const T a[6] = {0, 0, act4[0], act4[1], act4[2], act4[3]};
    const T ff = T(0.733) * T(10) / T(25);
    KS_UNROLL
    for (int i = 0; i < 3; i++) {
        T stuff = T3[3 * i + 2] * ff;
        T slide = T3[3 * i] * a[0] + T3[3 * i + 1] * a[1] + T3[3 * i + 2] * a[2];
        if (i < 2) { stuff = -stuff; slide = -slide; }
        ctrl[2 * i] = slide;
        ctrl[2 * i + 1] = stuff;
        ctrl[6 + i] = a[3 + i];
    }
}

// The termination test of build_obs alone (object centre at / above the 0.2 m target, ENV:631-687): the same operations
// on the same inputs, so the same answer - callers that must know where the observation goes before they build it.
template <typename T, typename C>
KS_HD bool object_lifted(const Model<T>& m, C snap) {
    T Ro[9], po[3], ow[3], t[3];
    snap_body<T>(snap, 9, Ro, po);
    mulRv(t, Ro, m.geom_pos[8]);
    add3(ow, po, t);
    const T target = T(0.2);
    return (kabs(ow[2] - target) < T(0.005)) || (ow[2] >= target);
}

// Full local observation + reward from a snapshot and the 17 ray distances.
// obs is written through `put(j, value)`.
template <typename T, typename C, typename Put>
KS_HD void build_obs(const Model<T>& m, C snap, const T* rays, Put put, T& reward, bool& lifted, T* info, int part = -1) {
    // part: -1 everything; 0..3 one quarter of the slots (the stepping kernel's tail gives the four waves of a workgroup one
    // part each: 0 finger geometry (slots 0-17, 73-80), 1 / 2 the finger-site distances (36-41 / 42-47), 3 the rest + reward)
    const bool all = part < 0, p0 = all || part == 0, p1 = all || part == 1, p2 = all || part == 2, p3 = all || part == 3;
    T R7[9], p7[3], Rpalm[9], ppalm[3], t[3];
    snap_body<T>(snap, 2, R7, p7);
    mulRR(Rpalm, R7, m.geom_R[1]);
    mulRv(t, R7, m.geom_pos[1]);
    add3(ppalm, p7, t);
    T T3[9], wrist[3];
    palm_transform(Rpalm, ppalm, T3, wrist);
    // finger link geom centres, order f1_prox f2_prox f3_prox f1_dist f2_dist f3_dist (ENV:478)
    T fw[6][3] = {}, fl[18] = {};
    KS_UNROLL
    for (int kq = 0; kq < 6 && p0; kq++) {
        const int g = kq < 3 ? 2 + 2 * kq : 3 + 2 * (kq - 3);
        T R[9], p[3];
        snap_body<T>(snap, m.geom_body[g], R, p);
        mulRv(t, R, m.geom_pos[g]);
        add3(fw[kq], p, t);
        T d[3];
        sub3(d, fw[kq], wrist);
        mulRv(&fl[3 * kq], T3, d);
    }
    T Ro[9], po[3], ow[3], ol[3];
    snap_body<T>(snap, 9, Ro, po);
    mulRv(t, Ro, m.geom_pos[8]);
    add3(ow, po, t);
    sub3(t, ow, wrist);
    mulRv(ol, T3, t);
    if (p0) {
        KS_UNROLL
        for (int j = 0; j < 18; j++) put(j, fl[j]);
    }
    if (p3) {
        // wrist in its own frame: T3 (wrist - wrist) = 0 (ENV:513-516)
        put(18, T(0)); put(19, T(0)); put(20, T(0));
        put(21, ol[0]); put(22, ol[1]); put(23, ol[2]);
        // joint states: jointpos sensors with the first two negated (ENV:356-362)
        KS_UNROLL
        for (int j = 0; j < 9; j++) {
            T v = snap(SNAP_JPOS + j);
            put(24 + j, j < 2 ? -v : v);
        }
        put(33, m.obj_size_obs[0]); put(34, m.obj_size_obs[1]); put(35, m.obj_size_obs[2]);
    }
    // finger-site to object distances (ENV:538-548), site order f1_prox f1_prox_1 f2_prox ... f1_dist ...
    {
        const int order[12] = {5, 6, 9, 10, 13, 14, 7, 8, 11, 12, 15, 16};
        KS_UNROLL
        for (int kq = 0; kq < 12; kq++) {
            if (!(kq < 6 ? p1 : p2)) continue;
            const int si = order[kq];
            T R[9], p[3], sp[3], d[3];
            snap_body<T>(snap, m.site_body[si], R, p);
            mulRv(t, R, m.site_pos[si]);
            add3(sp, p, t);
            sub3(d, sp, ow);
            put(36 + kq, norm3(d));
        }
    }
    // x / z angles (ENV:563-582)
    if (p3) {
        // arccos(y / sqrt(y^2 + s^2)) == atan2(|s|, y): identical value, but well conditioned in
        // fp32 when the object sits on the palm centre line (ratio -> 1, SURVEY O2 caution)
        T za = katan2(kabs(ol[0]), ol[1]);
        T xa = katan2(kabs(ol[2]), ol[1]);
        put(48, xa); put(49, za);
    }
    T rng[NRAY];
    KS_UNROLL
    for (int i = 0; i < NRAY; i++) { rng[i] = rays[i] == T(-1) ? T(6) : rays[i]; if (p3) put(50 + i, rng[i]); }
    const T g[3] = {-T3[2], -T3[5], -T3[8]};
    if (p3) { put(67, g[0]); put(68, g[1]); put(69, g[2]); }
    // experimental_sensor (ENV:290-343)
    if (p3) {
        T sx = 0, sy = 0, sz = 0, nh = 0;
        KS_UNROLL
        for (int i = 0; i < 5; i++) {
            if (rng[i] < T(0.06)) {
                T sp[3], d[3], l[3];
                mulRv(t, R7, m.site_pos[i]);
                add3(sp, p7, t);
                sub3(d, sp, wrist);
                mulRv(l, T3, d);
                l[1] += rng[i];
                sx += l[0]; sy += l[1]; sz += l[2];
                nh += 1;
            }
        }
        if (nh == 0) { put(70, T(0.2)); put(71, T(0.2)); put(72, T(0.2)); }
        else { put(70, sx / nh); put(71, sy / nh); put(72, sz / nh); }
    }
    if (p0) {
        T s1[3], s2[3];
        KS_UNROLL
        for (int i = 0; i < 3; i++) { s1[i] = fl[i] - fl[6 + i]; s2[i] = fl[i] - fl[3 + i]; }
        T front_area = tri_area(s1, s2);
        T top1 = tri_area(&fl[0], &fl[9]), top2 = tri_area(&fl[9], &fl[12]), top3 = tri_area(&fl[3], &fl[12]);
        T top4 = tri_area(&fl[6], &fl[15]), top5 = tri_area(&fl[9], &fl[15]);
        T total1 = top1 + top2 + top3, total2 = top1 + top4 + top5, top_area = total1 > total2 ? total1 : total2;
        const T z0 = m.obj_size_obs[0], z1 = m.obj_size_obs[1], z2 = m.obj_size_obs[2] * T(0.5);
        int am = 0;
        if (kabs(g[1]) > kabs(g[am])) am = 1;
        if (kabs(g[2]) > kabs(g[am])) am = 2;
        T fp, tp;
        if (am == 2) { fp = kabs(z0 * z2) / front_area; tp = kabs(z0 * z1) / top_area; }
        else if (am == 1) { fp = kabs(z0 * z2) / front_area; tp = kabs(z1 * z2) / top_area; }
        else { fp = kabs(z0 * z1) / front_area; tp = kabs(z0 * z2) / top_area; }
        put(73, fp); put(74, tp);
        KS_UNROLL
        for (int kq = 0; kq < 6; kq++) put(75 + kq, dot_product20(fw[kq], p7));
    }
    if (p3) put(81, dot_product20(ow, p7));
    // reward / termination (ENV:631-687, with_grasp_reward False)
    const T target = T(0.2);
    lifted = (kabs(ow[2] - target) < T(0.005)) || (ow[2] >= target);
    T lift = lifted ? T(50) : T(0);
    info[0] = 0; info[1] = 0; info[2] = lift;
    reward = lift;
}

}  // namespace ks
