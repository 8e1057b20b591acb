// ks_model_host.h -- host-side parser of the KSMB model blob (kinovagrasping_amd/model_compiler.py)
// into ks::Model<T>.  Header-only; used by the HIP library (device upload) and by the CPU lane check.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "ks_model.h"

namespace ks {

struct BlobRec {
    const unsigned char* data;
    uint32_t code, count, shape[4];
};

inline bool blob_find(const unsigned char* blob, size_t n, const char* name, BlobRec& out) {
    if (n < 16 || std::memcmp(blob, "KSMB", 4) != 0) return false;
    uint32_t ver, nrec;
    std::memcpy(&ver, blob + 4, 4);
    std::memcpy(&nrec, blob + 8, 4);
    if (ver != 5) return false;
    size_t off = 16;
    for (uint32_t i = 0; i < nrec && off + 48 <= n; i++) {
        char nm[25] = {0};
        std::memcpy(nm, blob + off, 24);
        uint32_t hdr[6];
        std::memcpy(hdr, blob + off + 24, 24);
        size_t isz = hdr[0] == 0 ? 8 : 4, bytes = (size_t)hdr[1] * isz;   // 0 = f64, 1 = i32, 2 = f32
        bytes += (8 - bytes % 8) % 8;
        if (off + 48 + bytes > n) return false;
        if (std::strcmp(nm, name) == 0) {
            out.data = blob + off + 48;
            out.code = hdr[0]; out.count = hdr[1];
            for (int k = 0; k < 4; k++) out.shape[k] = hdr[2 + k];
            return true;
        }
        off += 48 + bytes;
    }
    return false;
}

template <typename T> struct HostModel {
    Model<T> m;
    std::vector<T> vert[NMESH];
    std::vector<float> tri[NMESH], bvh_box[NMESH];
    std::vector<int> bvh_lr[NMESH];
    std::vector<unsigned short> adj_off[NMESH], adj[NMESH];
    std::vector<unsigned short> dirtab;     // [nmesh][SUPPORT_CELLS]
    std::string error;
};

namespace detail {
template <typename T> bool get_f64(const unsigned char* b, size_t n, const char* name, T* dst, size_t count, std::string& err) {
    BlobRec r;
    if (!blob_find(b, n, name, r) || r.code != 0 || r.count != count) { err = std::string("bad or missing record '") + name + "'"; return false; }
    for (size_t i = 0; i < count; i++) { double v; std::memcpy(&v, r.data + 8 * i, 8); dst[i] = (T)v; }
    return true;
}
inline bool get_i32(const unsigned char* b, size_t n, const char* name, int* dst, size_t count, std::string& err) {
    BlobRec r;
    if (!blob_find(b, n, name, r) || r.code != 1 || r.count != count) { err = std::string("bad or missing record '") + name + "'"; return false; }
    std::memcpy(dst, r.data, count * 4);
    return true;
}
inline void q2m(const double* q, double* R) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
}  // namespace detail

// Returns false and sets hm.error on a malformed blob or a model outside the supported topology.
template <typename T> bool parse_model(const void* vblob, size_t n, HostModel<T>& hm) {
    using namespace detail;
    const unsigned char* b = (const unsigned char*)vblob;
    Model<T>& m = hm.m;
    m.hull_pack = nullptr;       // filled by a device context (ks_api.hip, load_models)
    m.hull_pack_bytes = 0;
    std::string& e = hm.error;
    BlobRec probe;
    if (!blob_find(b, n, "opt", probe)) { e = "not a KSMB v5 model blob (merge <shape>.ksm with hand_raymesh.kst: model_compiler.load_model_blob)"; return false; }
    double opt[11], body_pos[30], body_quat[40], body_mass[10], body_ipos[30], body_iquat[40], body_inertia[30];
    double geom_pos[3 * NGEOM], geom_quat[4 * NGEOM], geom_size[3 * NGEOM], geom_rbound[NGEOM], geom_invw[NGEOM], site_pos[NSITE * 3], site_quat[NSITE * 4];
    double hl[6], binvw[20];
    bool ok = true;
    // geoms: the fixed nine, or - multi-geom objects, KS_MULTI_GEOM builds only - `object` plus its welded pieces (all of body 9)
    if (!blob_find(b, n, "geom_body", probe) || probe.code != 1 || probe.count < 9) { e = "bad or missing record 'geom_body'"; return false; }
    if (probe.count > (uint32_t)NGEOM) {
        e = MULTI_GEOM ? "more object geoms than this build holds" : "a multi-geom object (welded pieces) needs the multi-geom build of the library (libkinova_sim_mg.so)";
        return false;
    }
    const size_t ng = probe.count;
    m.ngeom = (int)ng;
    m.nmesh = (int)ng - 5;
    ok = ok && get_f64(b, n, "opt", opt, 11, e) && get_f64(b, n, "body_pos", body_pos, 30, e) && get_f64(b, n, "body_quat", body_quat, 40, e);
    ok = ok && get_f64(b, n, "body_mass", body_mass, 10, e) && get_f64(b, n, "body_ipos", body_ipos, 30, e);
    ok = ok && get_f64(b, n, "body_iquat", body_iquat, 40, e) && get_f64(b, n, "body_inertia", body_inertia, 30, e);
    ok = ok && get_f64(b, n, "slide_axis", &m.slide_axis[0][0], 9, e) && get_f64(b, n, "slide_range", &m.slide_range[0][0], 6, e);
    ok = ok && get_f64(b, n, "hinge_range", &m.hinge_range[0][0], 12, e) && get_f64(b, n, "hinge_limited", hl, 6, e);
    ok = ok && get_f64(b, n, "dof_damping", m.damping, NV, e) && get_f64(b, n, "dof_armature", m.armature, NV, e);
    ok = ok && get_f64(b, n, "geom_pos", geom_pos, 3 * ng, e) && get_f64(b, n, "geom_quat", geom_quat, 4 * ng, e);
    ok = ok && get_f64(b, n, "geom_size", geom_size, 3 * ng, e) && get_f64(b, n, "geom_rbound", geom_rbound, ng, e);
    ok = ok && get_i32(b, n, "geom_body", m.geom_body, ng, e) && get_i32(b, n, "geom_mesh", m.geom_mesh, ng, e);
    ok = ok && get_f64(b, n, "site_pos", site_pos, NSITE * 3, e) && get_f64(b, n, "site_quat", site_quat, NSITE * 4, e);
    ok = ok && get_i32(b, n, "site_body", m.site_body, NSITE, e);
    ok = ok && get_f64(b, n, "tendon_coef", &m.tendon_coef[0][0], 6, e) && get_f64(b, n, "actuator", m.act, 5, e);
    ok = ok && get_f64(b, n, "dof_invweight0", m.dof_invw, NV, e) && get_f64(b, n, "body_invweight0", binvw, 20, e);
    ok = ok && get_f64(b, n, "tendon_invweight0", m.tendon_invw, 3, e) && get_f64(b, n, "obj_size_obs", m.obj_size_obs, 3, e);
    if (!ok) return false;
    for (int g = (int)ng; g < NGEOM; g++) { m.geom_body[g] = 0; m.geom_mesh[g] = 0; }
    for (int g = 0; g < (int)ng; g++) {
        const bool obj = g >= OBJ_GEOM;
        if (m.geom_body[g] != (g == 0 ? 0 : (obj ? 9 : g + 1)) || m.geom_mesh[g] != (g == 0 ? -1 : (obj ? g - 5 : (g == 1 ? 0 : 1 + (g & 1))))) {
            e = "geom_body / geom_mesh outside the supported topology"; return false;
        }
        geom_invw[g] = binvw[2 * m.geom_body[g]];
    }
    if (blob_find(b, n, "geom_invweight0", probe) && !get_f64(b, n, "geom_invweight0", geom_invw, ng, e)) return false;
    m.dt = (T)opt[0]; m.impratio = (T)opt[1]; m.gravity_z = (T)opt[2];
    const double margin = opt[3];
    double tc = opt[4] < 2 * opt[0] ? 2 * opt[0] : opt[4], dr = opt[5], dmax = opt[7];
    m.solref_k = (T)(1.0 / (dmax * dmax * tc * tc * dr * dr));
    m.solref_b = (T)(2.0 / (dmax * tc));
    m.solimp[0] = (T)opt[6]; m.solimp[1] = (T)opt[7]; m.solimp[2] = (T)opt[8];
    m.mpr_tol = (T)opt[9]; m.mpr_iters = (int)opt[10];
    for (int i = 0; i < 3; i++) m.l7_pos[i] = (T)body_pos[2 * 3 + i];
    for (int i = 0; i < 6; i++) m.hinge_limited[i] = hl[i] != 0.0;
    for (int bi = 0; bi < NBODY; bi++) {
        m.mass[bi] = (T)body_mass[bi];
        m.body_invw[bi] = (T)binvw[2 * bi];
        for (int i = 0; i < 3; i++) m.ipos[bi][i] = (T)body_ipos[3 * bi + i];
        // inertia about the body z axis: (iR^T e_z) . diag . (iR^T e_z)
        double iR[9];
        q2m(&body_iquat[4 * bi], iR);
        double izz = 0;
        for (int k = 0; k < 3; k++) izz += iR[6 + k] * iR[6 + k] * body_inertia[3 * bi + k];
        m.izz[bi] = (T)izz;
        if (bi >= 3 && bi <= 8) {
            // closed-form finger dynamics need z to be a principal axis of the link inertia
            if (std::fabs(iR[8]) < 1 - 1e-9) { e = "finger link inertia frame is not aligned with the hinge axis"; return false; }
        }
        if (bi == 9) {
            double Ib[9];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) {
                    double v = 0;
                    for (int k = 0; k < 3; k++) v += iR[3 * r + k] * body_inertia[27 + k] * iR[3 * c + k];
                    Ib[3 * r + c] = v;
                }
            for (int k = 0; k < 9; k++) m.obj_Ib[k] = (T)Ib[k];
        }
    }
    for (int f = 0; f < 3; f++) {
        double R[9];
        const int bP = 3 + 2 * f, bD = 4 + 2 * f;
        q2m(&body_quat[4 * bP], R);
        for (int k = 0; k < 9; k++) m.fbase_R[f][k] = (T)R[k];
        q2m(&body_quat[4 * bD], R);
        // the planar-chain closed forms need the distal hinge parallel to the proximal one
        if (std::fabs(R[8]) < 1 - 1e-9) { e = "distal hinge axis is not parallel to the proximal hinge axis"; return false; }
        for (int k = 0; k < 9; k++) m.ftip_R[f][k] = (T)R[k];
        for (int k = 0; k < 3; k++) { m.fbase_pos[f][k] = (T)body_pos[3 * bP + k]; m.ftip_pos[f][k] = (T)body_pos[3 * bD + k]; }
    }
    for (int g = 0; g < NGEOM; g++) {
        double R[9];
        if (g >= (int)ng) {             // unused capacity: identity poses, nothing collides with them
            for (int k = 0; k < 9; k++) m.geom_R[g][k] = (T)(k % 4 == 0);
            for (int k = 0; k < 3; k++) { m.geom_pos[g][k] = T(0); m.geom_size[g][k] = T(0); }
            m.geom_rbound[g] = T(0);
            continue;
        }
        q2m(&geom_quat[4 * g], R);
        for (int k = 0; k < 9; k++) m.geom_R[g][k] = (T)R[k];
        for (int k = 0; k < 3; k++) { m.geom_pos[g][k] = (T)geom_pos[3 * g + k]; m.geom_size[g][k] = (T)geom_size[3 * g + k]; }
        m.geom_rbound[g] = (T)geom_rbound[g];
    }
    for (int s = 0; s < NSITE; s++) {
        double R[9];
        q2m(&site_quat[4 * s], R);
        for (int k = 0; k < 3; k++) { m.site_pos[s][k] = (T)site_pos[3 * s + k]; m.site_z[s][k] = (T)R[3 * k + 2]; }
    }
    BlobRec pr;
    if (!blob_find(b, n, "pairs", pr) || pr.shape[1] != 5 || pr.shape[0] > (uint32_t)NPAIR_MAX) { e = "bad 'pairs' record"; return false; }
    m.npair = (int)pr.shape[0];
    for (int p = m.npair; p < NPAIR_MAX; p++) { m.pair_g1[p] = m.pair_g2[p] = 0; m.pair_mu[p] = m.pair_margin[p] = m.pair_invw[p][0] = m.pair_invw[p][1] = T(0); }
    for (int p = 0; p < m.npair; p++) {
        double row[5];
        std::memcpy(row, pr.data + 40 * p, 40);
        m.pair_g1[p] = (int)row[0]; m.pair_g2[p] = (int)row[1];
        if (m.pair_g1[p] < 0 || m.pair_g1[p] >= m.pair_g2[p] || m.pair_g2[p] >= (int)ng) { e = "pair geoms out of range"; return false; }
        {
            const T w1 = (T)geom_invw[m.pair_g1[p]], w2 = (T)geom_invw[m.pair_g2[p]];
            const bool o1 = m.pair_g1[p] >= OBJ_GEOM, o2 = m.pair_g2[p] >= OBJ_GEOM;
            m.pair_invw[p][0] = (o1 ? T(0) : w1) + (o2 ? T(0) : w2);
            m.pair_invw[p][1] = (o1 ? w1 : T(0)) + (o2 ? w2 : T(0));
        }
        if (row[2] != row[3]) { e = "anisotropic pair friction is not supported"; return false; }
        if (!(row[4] >= 0) || row[4] > margin) { e = "pair margin outside [0, geom margin]"; return false; }
        m.pair_mu[p] = (T)row[2]; m.pair_margin[p] = (T)row[4];
    }
    model_pair_order(m);
    for (int s = m.nmesh; s < NMESH; s++) {
        m.mesh_nvert[s] = m.mesh_nvert_pad[s] = m.mesh_ntri[s] = m.mesh_nnode[s] = m.mesh_nchunk[s] = 0;
        m.mesh_vert[s] = nullptr; m.mesh_tri[s] = nullptr; m.mesh_bvh_box[s] = nullptr; m.mesh_bvh_lr[s] = nullptr;
        m.mesh_adj_off[s] = nullptr; m.mesh_adj[s] = nullptr;
    }
    for (int s = 0; s < m.nmesh; s++) {
        char nm[24];
        BlobRec r;
        std::snprintf(nm, sizeof nm, "mesh%d_vert", s);
        if (!blob_find(b, n, nm, r) || r.code != 0) { e = std::string("missing ") + nm; return false; }
        {
            const int nv = (int)r.shape[0], npad = (nv + 7) / 8 * 8;   // HULL_CHUNK
            // standard build: 10-bit vertex ids in the pair memory (ks_core.h PairWarm), 64 rounds of 16 in the plane scan's candidate mask
            if (nv < 1 || nv > HULL_VERT_MAX) { e = std::string(nm) + ": a hull has 1 .. " + std::to_string(HULL_VERT_MAX) + " vertices"; return false; }
            hm.vert[s].assign((size_t)npad * 4, T(0));
            for (int i = 0; i < npad; i++)
                for (int c = 0; c < 3; c++) { double v; std::memcpy(&v, r.data + 8 * (3 * (i < nv ? i : 0) + c), 8); hm.vert[s][4 * i + c] = (T)v; }
            m.mesh_nvert[s] = nv;
            m.mesh_nvert_pad[s] = npad;
        }
        m.mesh_vert[s] = hm.vert[s].data();
        {
            BlobRec rt, rb, rl;
            std::snprintf(nm, sizeof nm, "mesh%d_tri", s);
            if (!blob_find(b, n, nm, rt) || rt.code != 2 || rt.shape[1] != 9) { e = std::string("missing ") + nm; return false; }
            std::snprintf(nm, sizeof nm, "mesh%d_bvh_box", s);
            if (!blob_find(b, n, nm, rb) || rb.code != 2 || rb.shape[1] != 6) { e = std::string("missing ") + nm; return false; }
            std::snprintf(nm, sizeof nm, "mesh%d_bvh_lr", s);
            if (!blob_find(b, n, nm, rl) || rl.code != 1 || rl.shape[0] != rb.shape[0]) { e = std::string("missing ") + nm; return false; }
            hm.tri[s].resize(rt.count); std::memcpy(hm.tri[s].data(), rt.data, rt.count * 4);
            hm.bvh_lr[s].resize(rl.count); std::memcpy(hm.bvh_lr[s].data(), rl.data, rl.count * 4);
            // 4-wide nodes for the traversal, collapsed from the binary hierarchy of the blob: a node's children are its
            // grandchildren (a child that is a leaf stays one slot).  One 128-byte record = one L2 line: 4 boxes (24 floats),
            // 4 child words (>= 0: wide node id; < 0: leaf, -(1 + first_triangle * 8 + count); RAY_EMPTY: unused slot whose
            // box is inverted so that it is always missed), 4 words of padding.  Half the levels of the binary tree means half
            // the dependent memory round trips per walk.
            {
                std::vector<float> nb(rb.count);
                std::memcpy(nb.data(), rb.data, rb.count * 4);
                const int nn = (int)(rl.count / 2);
                const std::vector<int>& lr = hm.bvh_lr[s];
                for (int i = 0; i < nn; i++) {
                    const int ca = lr[2 * i], cb = lr[2 * i + 1];
                    if (cb >= 0 && (ca <= i || cb <= i || ca >= nn || cb >= nn)) { hm.error = "ray hierarchy: child ids out of order"; return false; }
                    if (cb < 0 && (-cb > 7 || ca < 0)) { hm.error = "ray hierarchy: leaf with more than 7 triangles"; return false; }
                }
                std::vector<float>& W = hm.bvh_box[s];
                W.clear();
                struct Item { int bnode, wide, depth; };
                std::vector<Item> todo;
                auto new_wide = [&]() { const int id = (int)(W.size() / 32); W.resize(W.size() + 32, 0.0f); return id; };
                int deepest = 0;
                if (nn > 0) { todo.push_back({0, new_wide(), 1}); }
                while (!todo.empty()) {
                    const Item it = todo.back();
                    todo.pop_back();
                    if (it.depth > deepest) deepest = it.depth;
                    int kids[4], nk = 0;
                    if (lr[2 * it.bnode + 1] < 0) kids[nk++] = it.bnode;                 // the whole mesh is one leaf
                    else {
                        for (int c = 0; c < 2; c++) {
                            const int ch = lr[2 * it.bnode + c];
                            if (lr[2 * ch + 1] < 0) kids[nk++] = ch;
                            else { kids[nk++] = lr[2 * ch]; kids[nk++] = lr[2 * ch + 1]; }
                        }
                    }
                    for (int k = 0; k < 4; k++) {
                        float box[6] = {1.0f, 1.0f, 1.0f, -1.0f, -1.0f, -1.0f};
                        int word = RAY_EMPTY;
                        if (k < nk) {
                            const int ch = kids[k];
                            std::memcpy(box, &nb[6 * (size_t)ch], 24);
                            if (lr[2 * ch + 1] < 0) word = -(1 + lr[2 * ch] * 8 + (-lr[2 * ch + 1]));
                            else { word = new_wide(); todo.push_back({ch, word, it.depth + 1}); }
                        }
                        float* w = &W[(size_t)it.wide * 32];             // (W may have been reallocated by new_wide)
                        std::memcpy(w + 6 * k, box, 24);
                        std::memcpy(w + 24 + k, &word, 4);
                    }
                }
                // limits of the traversal (ks_obs.h): 16-bit node ids on the stack, at most 3 pending children per level
                if (W.size() / 32 > 65535 || 3 * deepest > RAY_STACK) { hm.error = "ray hierarchy too large (wide nodes > 65535 or depth > RAY_STACK / 3)"; return false; }
            }
            m.mesh_ntri[s] = (int)rt.shape[0];
            m.mesh_nnode[s] = (int)rb.shape[0];
            m.mesh_tri[s] = hm.tri[s].data();
            m.mesh_bvh_box[s] = hm.bvh_box[s].data();
            m.mesh_bvh_lr[s] = hm.bvh_lr[s].data();
        }
        {
            BlobRec ro, ra;
            std::snprintf(nm, sizeof nm, "mesh%d_adj_off", s);
            if (!blob_find(b, n, nm, ro) || ro.code != 1 || (int)ro.count != m.mesh_nvert[s] + 1) { e = std::string("missing ") + nm; return false; }
            std::snprintf(nm, sizeof nm, "mesh%d_adj", s);
            if (!blob_find(b, n, nm, ra) || ra.code != 1) { e = std::string("missing ") + nm; return false; }
            std::vector<int> off(ro.count), adj(ra.count);
            std::memcpy(off.data(), ro.data, ro.count * 4);
            std::memcpy(adj.data(), ra.data, ra.count * 4);
            if ((int)ra.count != off.back() || m.mesh_nvert[s] > 65535) { e = std::string("bad adjacency ") + nm; return false; }
            hm.adj_off[s].assign(1, 0);
            hm.adj[s].clear();
            for (int v = 0; v < m.mesh_nvert[s]; v++) {
                for (int k = off[v]; k < off[v + 1]; k++) hm.adj[s].push_back((unsigned short)adj[k]);
                while (hm.adj[s].size() % 4) hm.adj[s].push_back((unsigned short)v);          // pad with self
                hm.adj_off[s].push_back((unsigned short)(hm.adj[s].size() / 4));
            }
            if (hm.adj[s].size() / 4 > 65535) { e = "hull graph too large for 16-bit chunk offsets"; return false; }
            while (hm.adj_off[s].size() % 4) hm.adj_off[s].push_back(hm.adj_off[s].back());   // whole 8-byte words (the stepping kernel stages them as such)
            m.mesh_nchunk[s] = (int)(hm.adj[s].size() / 4);
        }
        m.mesh_adj_off[s] = hm.adj_off[s].data();
        m.mesh_adj[s] = hm.adj[s].data();
    }
    // support vertex of the centre direction of every cube-map cell (exhaustive scans, once per model)
    hm.dirtab.assign((size_t)m.nmesh * SUPPORT_CELLS, 0);
    for (int s = 0; s < m.nmesh; s++)
        for (int c = 0; c < SUPPORT_CELLS; c++) {
            const int face = c / (SUPPORT_R * SUPPORT_R), iu = (c / SUPPORT_R) % SUPPORT_R, iv = c % SUPPORT_R, axis = face >> 1;
            double dir[3];
            dir[axis] = (face & 1) ? -1.0 : 1.0;
            dir[(axis + 1) % 3] = (iu + 0.5) * 2.0 / SUPPORT_R - 1.0;
            dir[(axis + 2) % 3] = (iv + 0.5) * 2.0 / SUPPORT_R - 1.0;
            double best = -1e300;
            int arg = 0;
            for (int v = 0; v < m.mesh_nvert[s]; v++) {
                const T* p = &hm.vert[s][4 * v];
                const double val = dir[0] * (double)p[0] + dir[1] * (double)p[1] + dir[2] * (double)p[2];
                if (val > best) { best = val; arg = v; }
            }
            hm.dirtab[s * SUPPORT_CELLS + c] = (unsigned short)arg;
        }
    m.mesh_dirtab = hm.dirtab.data();
    return true;
}

}  // namespace ks
