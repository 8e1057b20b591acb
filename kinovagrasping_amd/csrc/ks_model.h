// ks_model.h -- device-side model constants (one per loaded .ksm blob), templated on real type.
// Filled on the host by ks_model_host.cpp from the blob written by model_compiler.py; lives in
// device memory and is read through wave-uniform (scalar) loads by every kernel.
#pragma once
#include "ks_math.h"

namespace ks {

constexpr int NQ = 16, NV = 15, NU = 9, NBODY = 10, NGEOM = 9, NSITE = 17, NSENSOR = 26, NOBS = 82;
constexpr int NPAIR_MAX = 32;
constexpr int RAY_STACK = 24;   // pending-node bound of the ray-casting hierarchies (ks_obs.h: ray_mesh), checked at load
constexpr int RAY_EMPTY = (int)0x80000000;   // unused child slot of a 4-wide node
constexpr int NCON_MAX = 24;   // contacts kept per env per substep (oracle: KO_NCON_MAX)
constexpr int NRAY = 17;
#ifndef KS_SUPPORT_R
#define KS_SUPPORT_R 64
#endif
constexpr int SUPPORT_R = KS_SUPPORT_R, SUPPORT_CELLS = 6 * SUPPORT_R * SUPPORT_R;   // cube-map resolution of the support start tables

// status bits reported per env
constexpr int ST_CONTACT_OVERFLOW = 1, ST_NONFINITE = 2, ST_RAY_POOL_TIMEOUT = 4, ST_NEWTON_CAP = 8;

template <typename T> struct Model {
    T dt, impratio, gravity_z, mpr_tol;
    T solref_k, solref_b;      // k = 1/(dmax^2 tc^2 dr^2), b = 2/(dmax tc), tc >= 2 dt
    T solimp[3];
    int mpr_iters;
    T l7_pos[3];
    T slide_axis[3][3], slide_range[3][2];
    T hinge_range[6][2];
    int hinge_limited[6];
    T damping[NV], armature[NV];
    T mass[NBODY], ipos[NBODY][3];
    T izz[NBODY];              // inertia about the body z axis (hinge axis) for finger links
    T fbase_pos[3][3], fbase_R[3][9];   // proximal body frames in link_7
    T ftip_pos[3][3], ftip_R[3][9];     // distal body frames in the proximal
    T obj_Ib[9];               // object inertia about its COM, body frame (row-major)
    T geom_pos[NGEOM][3], geom_R[NGEOM][9], geom_rbound[NGEOM], geom_size[NGEOM][3];
    int geom_body[NGEOM], geom_mesh[NGEOM];
    T site_pos[NSITE][3], site_z[NSITE][3];
    int site_body[NSITE];
    int npair, pair_g1[NPAIR_MAX], pair_g2[NPAIR_MAX];
    T pair_mu[NPAIR_MAX], pair_margin[NPAIR_MAX];
    T tendon_coef[3][2];
    T act[5];                  // kv_slide, gear_motor, ctrlrange_slide, kv_finger, ctrlrange_finger
    T dof_invw[NV], body_invw[NBODY], tendon_invw[3];
    T obj_size_obs[3];
    int mesh_nvert[4], mesh_nvert_pad[4], mesh_ntri[4], mesh_nnode[4];
    const T* mesh_vert[4];     // [nvert_pad][4] (x,y,z,0) in the geom frame; rows >= nvert repeat vertex 0
    // rangefinder geometry: the original mesh triangles (geom frame) under a bounding-volume hierarchy
    const float* mesh_tri[4];      // [ntri][9], leaf ranges contiguous
    const float* mesh_bvh_box[4];  // [wide nodes][32] 4-wide ray hierarchy: 4 child boxes (min xyz, max xyz), 4 child words, padding (ks_model_host.h)
                                   // (int bits; leaf: first triangle, -count), 2 pad
    const int* mesh_bvh_lr[4];     // [nnode][2] internal: (left, right); leaf: (first triangle, -count)
    // vertex adjacency of the hull graph in chunks of 4 neighbour ids (uint16, ascending, padded with the
    // vertex itself, which never wins a strict comparison): one 8-byte read yields four neighbours
    int mesh_nchunk[4];
    const unsigned short* mesh_adj_off[4];  // [nvert+1] first chunk of every vertex
    const unsigned short* mesh_adj[4];      // [nchunk][4]
    // support vertex of every hull for the centre direction of every cell of a cube map (6 faces x R x R): where a
    // hill climb towards an arbitrary direction starts, [4][SUPPORT_CELLS]; global memory (12 KB, L1/L2 resident)
    const unsigned short* mesh_dirtab;
    // the hull tables once more as ONE block in the order the stepping kernels keep them in LDS ([vert 0..3][adj_off 0 | adj 0 | ..
    // | adj_off 3 | adj 3], padded to 16 bytes): staged with a single copy (ks_api.hip, stage_tables); device contexts only
    const void* hull_pack;
    int hull_pack_bytes;
};

}  // namespace ks
