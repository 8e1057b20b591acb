// ks_model.h -- device-side model constants (one per loaded .ksm blob), templated on real type.
// Filled on the host by ks_model_host.cpp from the blob written by model_compiler.py; lives in
// device memory and is read through wave-uniform (scalar) loads by every kernel.
#pragma once
#include "ks_math.h"

namespace ks {

constexpr int NQ = 16, NV = 15, NU = 9, NBODY = 10, NSITE = 17, NSENSOR = 26, NOBS = 82;
// Collision topology CAPACITIES of this build (a model brings its counts: Model::ngeom, npair, nmesh).
//  standard   : ground + 7 hand geoms + `object`; 8 explicit + 22 dynamic pairs; hulls of palm / proximal / distal / object.
//  multi-geom : (-DKS_MULTI_GEOM, libkinova_sim_mg.so) `object` + up to 8 jointless child bodies welded to it, one mesh geom each - the
//               reference's Bottle / TBottle / Bowl / RBowl models (kinova_description/..._sbottle.xml:158-186): 17 geoms, 30 + 8 x 8 pairs,
//               3 + 9 hulls of up to 4096 vertices (hull tables in global memory, so it also takes single-geom objects whose hull exceeds the standard build's 1024: Lemon).  The welded pieces are geoms of body 9 (one rigid body, composite inertial).
#ifdef KS_MULTI_GEOM
constexpr bool MULTI_GEOM = true;
constexpr int NGEOM = 17, NPAIR_MAX = 96, NMESH = 12, HULL_VERT_MAX = 4096;   // (4096: the Lemon stand-in's hull has 2434 vertices, the short bottle's base 1546)
#else
constexpr bool MULTI_GEOM = false;
constexpr int NGEOM = 9, NPAIR_MAX = 32, NMESH = 4, HULL_VERT_MAX = 1024;
#endif
constexpr int OBJ_GEOM = 8;     // the geom named `object` (pieces of a multi-geom object follow it)
// contacts kept per env per substep, in pair order (oracle: ko_sim.ncon_max, same values): 24 for the nine-geom models; 40 for the
// multi-geom objects (a bowl lying on the floor rests on up to five pieces x 4 plane contacts before a finger touches it)
#ifndef KS_MG_NCON
#define KS_MG_NCON 40
#endif
constexpr int NCON_MAX = MULTI_GEOM ? KS_MG_NCON : 24;
constexpr int RAY_STACK = 24;   // pending-node bound of the ray-casting hierarchies (ks_obs.h: ray_mesh), checked at load
constexpr int RAY_EMPTY = (int)0x80000000;   // unused child slot of a 4-wide node
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
    int ngeom, nmesh;          // counts of this model (<= NGEOM, NMESH)
    int npair, pair_g1[NPAIR_MAX], pair_g2[NPAIR_MAX];
    // the pairs as the stepping kernels share them out (filled at load, model_pair_order): plane pairs in index order; hull pairs by how often
    // they get past their culls - the ones that hardly ever do last, so that a lane that owns two hull pairs owns at most one busy one
    int nplane, nhull;
    unsigned char plane_order[NPAIR_MAX], hull_order[NPAIR_MAX];
    T pair_mu[NPAIR_MAX], pair_margin[NPAIR_MAX];
    // translational inverse weights behind a contact's regularisation (MuJoCo: body_invweight0 of the two bodies that own the geoms),
    // per pair: [0] the hand / ground geom's share, [1] the object geoms' share (scaled with a per-env object mass).  A piece of a
    // multi-geom object is a body of its own in MuJoCo and brings its own value (blob record geom_invweight0).
    T pair_invw[NPAIR_MAX][2];
    T tendon_coef[3][2];
    T act[5];                  // kv_slide, gear_motor, ctrlrange_slide, kv_finger, ctrlrange_finger
    T dof_invw[NV], body_invw[NBODY], tendon_invw[3];
    T obj_size_obs[3];
    int mesh_nvert[NMESH], mesh_nvert_pad[NMESH], mesh_ntri[NMESH], mesh_nnode[NMESH];
    const T* mesh_vert[NMESH]; // [nvert_pad][4] (x,y,z,0) in the geom frame; rows >= nvert repeat vertex 0
    // rangefinder geometry: the original mesh triangles (geom frame) under a bounding-volume hierarchy
    const float* mesh_tri[NMESH];      // [ntri][9], leaf ranges contiguous
    const float* mesh_bvh_box[NMESH];  // [wide nodes][32] 4-wide ray hierarchy: 4 child boxes (min xyz, max xyz), 4 child words, padding (ks_model_host.h)
                                   // (int bits; leaf: first triangle, -count), 2 pad
    const int* mesh_bvh_lr[NMESH]; // [nnode][2] internal: (left, right); leaf: (first triangle, -count)
    // vertex adjacency of the hull graph in chunks of 4 neighbour ids (uint16, ascending, padded with the
    // vertex itself, which never wins a strict comparison): one 8-byte read yields four neighbours
    int mesh_nchunk[NMESH];
    const unsigned short* mesh_adj_off[NMESH];  // [nvert+1] first chunk of every vertex
    const unsigned short* mesh_adj[NMESH];      // [nchunk][4]
    // support vertex of every hull for the centre direction of every cell of a cube map (6 faces x R x R): where a
    // hill climb towards an arbitrary direction starts, [nmesh][SUPPORT_CELLS]; global memory (48 KB per hull, L1/L2 resident)
    const unsigned short* mesh_dirtab;
    // the hull tables once more as ONE block in the order the stepping kernels keep them in LDS ([vert 0..3][adj_off 0 | adj 0 | ..
    // | adj_off 3 | adj 3], padded to 16 bytes): staged with a single copy (ks_api.hip, stage_tables); device contexts of the standard build only
    const void* hull_pack;
    int hull_pack_bytes;
};

// How often a hull pair of the HAND's own geoms (1 palm, 2 / 4 / 6 the proximal, 3 / 5 / 7 the distal links of fingers 1 - 3; finger 1 opposes 2 and 3)
// gets past its culls: 0 = often (the finger tips against each other, finger 1's proximal link against finger 2's tip), 1 = seldom, 2 = hardly ever
// (proximal links against each other, finger 1's tip against the others' proximal links).  Measured in the bench's regime
// (tools/experiments/hull_passes.py, profiles/r05_hull_passes.txt); everything else - the object's pairs - is class 0.
KS_HD int hull_pair_rarity(int g1, int g2) {
    constexpr unsigned long long bit = 1ull;
    constexpr unsigned long long HARDLY = (bit << (2 * 8 + 4)) | (bit << (2 * 8 + 6)) | (bit << (4 * 8 + 6)) | (bit << (3 * 8 + 4)) | (bit << (3 * 8 + 6)) | (bit << (5 * 8 + 6));
    constexpr unsigned long long SELDOM = (bit << (1 * 8 + 3)) | (bit << (1 * 8 + 5)) | (bit << (1 * 8 + 7)) | (bit << (3 * 8 + 7)) | (bit << (2 * 8 + 7)) | (bit << (4 * 8 + 7));
    const bool hand = g1 >= 1 && g1 <= 7 && g2 >= 1 && g2 <= 7;
    const int key = hand ? g1 * 8 + g2 : 0;                      // (bit 0 of both masks is clear)
    return (int)((HARDLY >> key) & 1) * 2 + (int)((SELDOM >> key) & 1);
}
// The list of hull pairs is what the lanes of a team share out (lane k: list entries k, then - from the last lanes down - the entries beyond the first
// 16): a lane with TWO live pairs costs its whole wave a second narrow-phase pass (measured with the index-order list in the bench's regime: 1.76 passes per
// wave and substep, because the two busiest finger-tip pairs shared a lane; with this order 0.99).  The list therefore ends with the pairs that are hardly
// ever live and has the seldom ones in front of them, so that in the standard model (22 hull pairs) the six lanes that own two pairs own a seldom and a
// hardly-ever one.  Contacts are merged by pair INDEX: the order of this list changes no result.  Called once when a model is loaded.
template <typename T> inline void model_pair_order(Model<T>& m) {
    m.nplane = m.nhull = 0;
    for (int pi = 0; pi < m.npair; pi++)
        if (m.pair_g1[pi] == 0) m.plane_order[m.nplane++] = (unsigned char)pi;
    for (int cls = 0; cls < 3; cls++)
        for (int pi = 0; pi < m.npair; pi++)
            if (m.pair_g1[pi] != 0 && hull_pair_rarity(m.pair_g1[pi], m.pair_g2[pi]) == cls) m.hull_order[m.nhull++] = (unsigned char)pi;
    for (int k = m.nplane; k < NPAIR_MAX; k++) m.plane_order[k] = 0;
    for (int k = m.nhull; k < NPAIR_MAX; k++) m.hull_order[k] = 0;
}

}  // namespace ks
