/* oracle/ko_physics.c -- fp64 single-env restatement of mj_step (MuJoCo 1.50 semantics, Euler)
 * for the Kinova j2s7s300 end-effector model.  TEST INFRASTRUCTURE ONLY (see ko.h).
 *
 * Reference call site: kinova_gripper_env.py:1535 (`self._sim.step()`), :703/:353 (`forward`).
 * The arithmetic itself is third-party (MuJoCo 1.50, not vendored, not installable here): this
 * file restates MuJoCo's published pipeline (documentation, "Computation" chapter; SURVEY.md
 * Appendix B) stage by stage.  PINNED against every number of real MuJoCo 1.50 output the reference's authors recorded and left in their
 * tree (tests/golden/mujoco_recorded.npz, tests/test_mujoco_recorded.py, tests/old_env.py):
 *   - Old Code/Pose_file_2.csv, the one recorded CONTACT trajectory (63 rows x 48 columns, one row per env.step()): replayed free-running - only the four
 *     un-recorded commands of a row are recovered from its four actuated joint angles, the other 44 columns are predictions - rows 0-40 agree in all
 *     48 columns to 1.05e-10 (the recording's ten decimals): release inside the floor, a box pushed 5 cm on its edge with sliding friction, a grasp
 *     between two finger pads, the first 2 cm of lift; rows 41-45 (third finger arrives, lift to 0.10 m) to 8e-8; rows 46-62 (rest of the lift) to
 *     5.5e-4 (object 1.05e-4) - there finger 1's pad rolls over the box's vertical edge and MuJoCo's own choice among 4-way ties of its analytic box
 *     support decides the portal (docs/recorded_mujoco.md; a searched tie pattern reproduces rows 46-47 to 3e-8);
 *   - Old Code/Pose_file.csv: free-closing finger joint traces to 3.3e-5 rad (actuators, damping, armature, soft tendon equality, gravity, sensor lag);
 *   - expert_plots/: ten recorded demonstrations (8 of 10 outcomes, the 6 common lifts to the env-step) and the naive controller's success / failure maps.
 * So the kinematics, the legacy mesh inertia, the explicit pairs' margin 0, the operand order of the convex queries, the contact model (impedance,
 * pyramid rows), Newton and the Euler step are pinned by positions through 45 rows of contact, grasp and lift.  NOT pinned, because no MuJoCo datum
 * exists for them anywhere under /root/reference: contact FORCES and qvel as such, mesh objects other than the primitive box, the rotated / top hand
 * poses, and everything specific to the multi-geom objects (ko_model.c: PARITY UNPINNED for those).
 *
 * Deliberately written in the plainest dense form (15x15 matrices, full Jacobians) so that it
 * is an independent check of the specialised HIP kernels.
 *
 * Third-party algorithms restated here: the penetration query (mpr_penetration and its helpers portal_dir, expand_portal,
 * find_pos, portal_reach_tolerance, point-triangle distance) follows the structure of libccd's ccdMPRPenetration
 * (libccd, (c) Daniel Fiser, BSD-3-Clause - the collision library MuJoCo 1.50 links; it is NOT part of /root/reference and
 * none of its source text is used); the distance query is the textbook GJK (Gilbert, Johnson, Keerthi 1988).
 *
 *   S1 ko_kinematics      mj_kinematics / mj_comPos
 *   S2 mass_matrix        mj_crb + mj_factorM         (here: M = sum_b J_b^T I_b J_b, Cholesky)
 *   S3 bias/passive/act   mj_rne / mj_passive / mj_fwdActuation
 *   S4 collision          mj_collision                (plane-hull, hull-hull via MPR)
 *   S5 make_constraint    mj_makeConstraint / mj_projectConstraint / mj_referenceConstraint
 *   S6 solve_newton       mj_fwdConstraint (Newton on the primal problem, MuJoCo's default for this XML; solve_pgs =
 *                         projected Gauss-Seidel on the dual, kept as an independent cross-check of the same optimum:
 *                         tests/test_oracle_known_answers.py)
 *   S7 euler              mj_Euler (implicit joint damping)
 *   S8 sensors            mj_sensorPos (jointpos, rangefinder)
 */
#include "ko.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef KO_GJK_TOL
#define KO_GJK_TOL 1e-6
#endif
#define CCD_EPS 1e-15
#define PLANE_TIE_EPS 1e-12 /* plane-hull: vertices within this of the deepest one count as equally deep, lowest index first */
#ifdef KO_SUPPORT_SKEW_OVERRIDE
#define KO_SUPPORT_SKEW KO_SUPPORT_SKEW_OVERRIDE
#else
#define KO_SUPPORT_SKEW 1e-6
#endif
#define KO_SKEW_X 0.5377
#define KO_SKEW_Y (-0.6240)
#define KO_SKEW_Z 0.5671
#define PLANE_MESH_TOL 0.3 /* extra plane-hull contacts must be > 0.3*rbound apart */
#define MINVAL 1e-15

static const int body_parent[KO_NBODY] = {0, 0, 1, 2, 3, 2, 5, 2, 7, 0};

/* ------------------------------------------------------------------ small vector helpers */
static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(double *r, const double *a, const double *b) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    r[0] = x; r[1] = y; r[2] = z;
}
static void sub3(double *r, const double *a, const double *b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static void add3(double *r, const double *a, const double *b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
static void copy3(double *r, const double *a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static void scl3(double *r, const double *a, double s) { r[0] = a[0] * s; r[1] = a[1] * s; r[2] = a[2] * s; }
static void addscl3(double *r, const double *a, double s) { r[0] += a[0] * s; r[1] += a[1] * s; r[2] += a[2] * s; }
static double norm3(const double *a) { return sqrt(dot3(a, a)); }
static double normalize3(double *a) {
    double n = norm3(a);
    if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; return 0; }
    a[0] /= n; a[1] /= n; a[2] /= n;
    return n;
}
/* r = R * v (R row-major 3x3) */
static void mulmatvec3(double *r, const double *R, const double *v) {
    double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    double y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    double z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
    r[0] = x; r[1] = y; r[2] = z;
}
/* r = R^T * v */
static void mulmatTvec3(double *r, const double *R, const double *v) {
    double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
    double y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
    double z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
    r[0] = x; r[1] = y; r[2] = z;
}
static void mulmat3(double *r, const double *A, const double *B) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(r, t, sizeof t);
}
static void quat2mat(double *R, const double *q) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
static void quatmul(double *r, const double *a, const double *b) {
    double t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                   a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                   a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                   a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
    memcpy(r, t, sizeof t);
}
static void quatnormalize(double *q) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
    for (int i = 0; i < 4; i++) q[i] /= n;
}

/* ------------------------------------------------------------------ S1 kinematics */
void ko_kinematics(ko_sim *s) {
    const ko_model *m = s->m;
    double xquat[KO_NBODY][4];
    memset(s->xpos[0], 0, 3 * sizeof(double));
    xquat[0][0] = 1; xquat[0][1] = xquat[0][2] = xquat[0][3] = 0;
    quat2mat(s->xmat[0], xquat[0]);
    for (int b = 1; b < KO_NBODY; b++) {
        int p = body_parent[b];
        if (b == 9) { /* free joint: pose comes straight from qpos */
            copy3(s->xpos[b], &s->qpos[9]);
            memcpy(xquat[b], &s->qpos[12], 4 * sizeof(double));
            quatnormalize(xquat[b]);
        } else {
            double t[3];
            mulmatvec3(t, s->xmat[p], m->body_pos[b]);
            add3(s->xpos[b], s->xpos[p], t);
            /* link_7's orientation is the per-episode hand orientation (ENV:870-876) */
            quatmul(xquat[b], xquat[p], b == 2 ? s->hand_quat : m->body_quat[b]);
            quatnormalize(xquat[b]);
            if (b == 2) { /* three slides along body axes */
                double R[9];
                quat2mat(R, xquat[b]);
                for (int k = 0; k < 3; k++) {
                    double ax[3];
                    mulmatvec3(ax, R, m->slide_axis[k]);
                    addscl3(s->xpos[b], ax, s->qpos[k]);
                }
            } else if (b >= 3) { /* hinge about local z through the body origin */
                double a = s->qpos[b], qz[4] = {cos(0.5 * a), 0, 0, sin(0.5 * a)};
                quatmul(xquat[b], xquat[b], qz);
                quatnormalize(xquat[b]);
            }
        }
        quat2mat(s->xmat[b], xquat[b]);
        double t[3], Ri[9];
        mulmatvec3(t, s->xmat[b], m->body_ipos[b]);
        add3(s->xipos[b], s->xpos[b], t);
        quat2mat(Ri, m->body_iquat[b]);
        mulmat3(s->ximat[b], s->xmat[b], Ri);
    }
    for (int g = 0; g < m->ngeom; g++) {
        int b = m->geom_body[g];
        double t[3], Rg[9];
        mulmatvec3(t, s->xmat[b], m->geom_pos[g]);
        add3(s->geom_xpos[g], s->xpos[b], t);
        quat2mat(Rg, m->geom_quat[g]);
        mulmat3(s->geom_xmat[g], s->xmat[b], Rg);
    }
    for (int i = 0; i < KO_NSITE; i++) {
        int b = m->site_body[i];
        double t[3], Rs[9];
        mulmatvec3(t, s->xmat[b], m->site_pos[i]);
        add3(s->site_xpos[i], s->xpos[b], t);
        quat2mat(Rs, m->site_quat[i]);
        mulmat3(s->site_xmat[i], s->xmat[b], Rs);
    }
}

/* Jacobian of a world point attached to `body`: translational Jp[3][NV], rotational Jr[3][NV].
 * Free joint convention: qvel[9:12] world linear velocity of the body origin, qvel[12:15]
 * angular velocity in the body frame. */
static void jac_point(const ko_sim *s, int body, const double *x, double Jp[3][KO_NV], double Jr[3][KO_NV]) {
    const ko_model *m = s->m;
    memset(Jp, 0, 3 * KO_NV * sizeof(double));
    memset(Jr, 0, 3 * KO_NV * sizeof(double));
    for (int b = body; b != 0; b = body_parent[b]) {
        if (b == 2) {
            for (int k = 0; k < 3; k++) {
                double ax[3];
                mulmatvec3(ax, s->xmat[2], m->slide_axis[k]);
                for (int r = 0; r < 3; r++) Jp[r][k] = ax[r];
            }
        } else if (b >= 3 && b <= 8) {
            double z[3] = {s->xmat[b][2], s->xmat[b][5], s->xmat[b][8]}, r[3], c[3];
            sub3(r, x, s->xpos[b]);
            cross3(c, z, r);
            for (int i = 0; i < 3; i++) { Jr[i][b] = z[i]; Jp[i][b] = c[i]; }
        } else if (b == 9) {
            double r[3];
            sub3(r, x, s->xpos[9]);
            for (int k = 0; k < 3; k++) {
                Jp[k][9 + k] = 1.0;
                double ax[3] = {s->xmat[9][k], s->xmat[9][3 + k], s->xmat[9][6 + k]}, c[3];
                cross3(c, ax, r);
                for (int i = 0; i < 3; i++) { Jr[i][12 + k] = ax[i]; Jp[i][12 + k] = c[i]; }
            }
        }
    }
}

/* mass of body b and the factor on its inertia: the object's may be replaced per sim (config 5) */
static double body_mass_of(const ko_sim *s, int b) { return (b == KO_NBODY - 1 && s->obj_mass > 0) ? s->obj_mass : s->m->body_mass[b]; }
static double inertia_scale_of(const ko_sim *s, int b) { return body_mass_of(s, b) / s->m->body_mass[b]; }
/* translational inverse weight of the MuJoCo body that owns geom g (the pieces of a multi-geom object: their own, ko.h) */
static double geom_invweight_of(const ko_sim *s, int g) {
    const ko_model *m = s->m;
    const int b = m->geom_body[g];
    double w = m->geom_invweight0[g];
    if (b == KO_NBODY - 1 && s->obj_mass > 0) w *= (m->body_mass[b] + m->dof_armature[9]) / (s->obj_mass + m->dof_armature[9]);
    return w;
}

static void world_inertia(const ko_sim *s, int b, double Iw[9]) {
    const double *Ri = s->ximat[b], *d0 = s->m->body_inertia[b];
    const double sc = inertia_scale_of(s, b), d[3] = {d0[0] * sc, d0[1] * sc, d0[2] * sc};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            Iw[3 * i + j] = Ri[3 * i] * d[0] * Ri[3 * j] + Ri[3 * i + 1] * d[1] * Ri[3 * j + 1] + Ri[3 * i + 2] * d[2] * Ri[3 * j + 2];
}

/* ------------------------------------------------------------------ S2 mass matrix + Cholesky */
static void mass_matrix(ko_sim *s) {
    const ko_model *m = s->m;
    memset(s->M, 0, sizeof s->M);
    for (int i = 0; i < KO_NV; i++) s->M[i][i] = m->dof_armature[i];
    for (int b = 2; b < KO_NBODY; b++) {
        double Jp[3][KO_NV], Jr[3][KO_NV], Iw[9];
        jac_point(s, b, s->xipos[b], Jp, Jr);
        world_inertia(s, b, Iw);
        for (int i = 0; i < KO_NV; i++)
            for (int j = 0; j < KO_NV; j++) {
                double acc = 0;
                for (int r = 0; r < 3; r++) {
                    acc += body_mass_of(s, b) * Jp[r][i] * Jp[r][j];
                    acc += Jr[r][i] * (Iw[3 * r] * Jr[0][j] + Iw[3 * r + 1] * Jr[1][j] + Iw[3 * r + 2] * Jr[2][j]);
                }
                s->M[i][j] += acc;
            }
    }
}

/* dense Cholesky A = L L^T (lower) */
static void cholesky(const double A[KO_NV][KO_NV], double L[KO_NV][KO_NV]) {
    memset(L, 0, KO_NV * KO_NV * sizeof(double));
    for (int j = 0; j < KO_NV; j++) {
        double d = A[j][j];
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
        L[j][j] = sqrt(d > MINVAL ? d : MINVAL);
        for (int i = j + 1; i < KO_NV; i++) {
            double v = A[i][j];
            for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k];
            L[i][j] = v / L[j][j];
        }
    }
}
static void chol_solve(const double L[KO_NV][KO_NV], const double *b, double *x) {
    double y[KO_NV];
    for (int i = 0; i < KO_NV; i++) {
        double v = b[i];
        for (int k = 0; k < i; k++) v -= L[i][k] * y[k];
        y[i] = v / L[i][i];
    }
    for (int i = KO_NV - 1; i >= 0; i--) {
        double v = y[i];
        for (int k = i + 1; k < KO_NV; k++) v -= L[k][i] * x[k];
        x[i] = v / L[i][i];
    }
}

/* ------------------------------------------------------------------ S3 bias (RNE with qacc=0) */
static void bias_forces(ko_sim *s) {
    const ko_model *m = s->m;
    double w[KO_NBODY][3], alpha[KO_NBODY][3], a0[KO_NBODY][3]; /* angular vel, vel-product angular acc, origin acc */
    memset(w, 0, sizeof w); memset(alpha, 0, sizeof alpha); memset(a0, 0, sizeof a0);
    for (int b = 2; b < KO_NBODY; b++) {
        int p = body_parent[b];
        if (b == 9) {
            mulmatvec3(w[b], s->xmat[9], &s->qvel[12]); /* alpha, a0 stay 0 */
        } else if (b == 2) {
            /* slides on a non-rotating parent: no velocity-product acceleration */
            copy3(w[b], w[p]); copy3(alpha[b], alpha[p]); copy3(a0[b], a0[p]);
        } else {
            double z[3] = {s->xmat[b][2], s->xmat[b][5], s->xmat[b][8]}, zq[3], r[3], t[3], u[3];
            scl3(zq, z, s->qvel[b]);
            add3(w[b], w[p], zq);
            cross3(t, w[p], zq);
            add3(alpha[b], alpha[p], t);
            sub3(r, s->xpos[b], s->xpos[p]);
            cross3(t, alpha[p], r);
            cross3(u, w[p], r);
            cross3(u, w[p], u);
            add3(a0[b], a0[p], t);
            add3(a0[b], a0[b], u);
        }
    }
    memset(s->qfrc_bias, 0, sizeof s->qfrc_bias);
    for (int b = 2; b < KO_NBODY; b++) {
        double Jp[3][KO_NV], Jr[3][KO_NV], Iw[9], c[3], ac[3], t[3], u[3], F[3], T[3], Iwv[3];
        jac_point(s, b, s->xipos[b], Jp, Jr);
        world_inertia(s, b, Iw);
        sub3(c, s->xipos[b], s->xpos[b]);
        cross3(t, alpha[b], c);
        cross3(u, w[b], c);
        cross3(u, w[b], u);
        add3(ac, a0[b], t);
        add3(ac, ac, u);
        ac[2] -= m->gravity_z; /* a_com - g */
        scl3(F, ac, body_mass_of(s, b));
        mulmatvec3(T, Iw, alpha[b]);
        mulmatvec3(Iwv, Iw, w[b]);
        cross3(t, w[b], Iwv);
        add3(T, T, t);
        for (int i = 0; i < KO_NV; i++)
            for (int r = 0; r < 3; r++) s->qfrc_bias[i] += Jp[r][i] * F[r] + Jr[r][i] * T[r];
    }
}

static double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

static void passive_and_actuation(ko_sim *s) {
    const ko_model *m = s->m;
    for (int i = 0; i < KO_NV; i++) s->qfrc_passive[i] = -m->dof_damping[i] * s->qvel[i];
    memset(s->qfrc_actuator, 0, sizeof s->qfrc_actuator);
    double kvs = m->actuator[0], gear = m->actuator[1], rs = m->actuator[2], kvf = m->actuator[3], rf = m->actuator[4];
    for (int k = 0; k < 3; k++) {
        /* <velocity kv ctrllimited> : kv*(clamp(ctrl) - qvel) ; <motor gear> : gear*ctrl   (XML:214-219) */
        s->qfrc_actuator[k] = kvs * (clampd(s->ctrl[2 * k], -rs, rs) - s->qvel[k]) + gear * s->ctrl[2 * k + 1];
        /* finger proximal velocity servos (XML:220-222) */
        s->qfrc_actuator[3 + 2 * k] = kvf * (clampd(s->ctrl[6 + k], -rf, rf) - s->qvel[3 + 2 * k]);
    }
}

/* ------------------------------------------------------------------ S4 collision */
typedef struct { double v[3], v1[3], v2[3]; } supp_t;
typedef struct { ko_sim *s; int g1, g2; double half_margin; } mpr_ctx;

static void hull_support(const ko_sim *s, int g, const double *dir, double half_margin, double *out) {
    const ko_model *m = s->m;
    int mesh = m->geom_mesh[g], n = m->mesh_nvert[mesh], bi = 0;
    const double *V = m->mesh_vert[mesh];
    double ld[3], best = -1e300;
    mulmatTvec3(ld, s->geom_xmat[g], dir);
    /* Tie rule shared with the kernels (ks_core.h: pair_support).  MPR and GJK query supports along the normals of faces they
     * have just built from the hulls' own vertices: every vertex of such a face then attains the maximum to the last bit, rounding
     * decides which one "wins", and the portal / simplex path - hence the contact POINT and, on polytopes that approximate round
     * shapes, even the facet the normal is taken from - follows that coin.  The hull-frame direction is therefore skewed by a
     * fixed 1e-6 of its size before the scan: 1e-7 m of support error at most (a tenth of MPR's tolerance), but an order above the
     * rounding of an fp32 direction and ten above fp64's, so that the fp64 oracle, the fp64 kernels and - mostly - the fp32 product
     * take the same vertex.  Measured: with 1e-6 and 1e-5 the replay of the recorded MuJoCo 1.50 contact trajectory is unchanged
     * (rows 0-21 to 1.9e-10), with 1e-4 it leaves the recording at row 15; fp32 lane vs oracle on 330 CubeS grasp states: contact
     * points agree 0.994 -> 1.000, worst one-step |dqpos| 2.6e-4 -> 2.2e-5. */
    {
        const double sk = KO_SUPPORT_SKEW * (fabs(ld[0]) + fabs(ld[1]) + fabs(ld[2]));
        ld[0] += sk * KO_SKEW_X; ld[1] += sk * KO_SKEW_Y; ld[2] += sk * KO_SKEW_Z;
    }
    for (int i = 0; i < n; i++) {
        double d = V[3 * i] * ld[0] + V[3 * i + 1] * ld[1] + V[3 * i + 2] * ld[2];
        if (d > best) { best = d; bi = i; }
    }
    mulmatvec3(out, s->geom_xmat[g], &V[3 * bi]);
    add3(out, out, s->geom_xpos[g]);
    addscl3(out, dir, half_margin);
}

static void mpr_support(mpr_ctx *c, const double *dir, supp_t *o) {
    double nd[3] = {-dir[0], -dir[1], -dir[2]};
    hull_support(c->s, c->g1, dir, c->half_margin, o->v1);
    hull_support(c->s, c->g2, nd, c->half_margin, o->v2);
    sub3(o->v, o->v1, o->v2);
    c->s->mpr_support_calls++;
}

static int is_zero(double x) { return fabs(x) < CCD_EPS; }
static int vec_is_zero(const double *v) { return is_zero(v[0]) && is_zero(v[1]) && is_zero(v[2]); }

static void portal_dir(const supp_t *v1, const supp_t *v2, const supp_t *v3, double *dir) {
    double a[3], b[3];
    sub3(a, v2->v, v1->v);
    sub3(b, v3->v, v1->v);
    cross3(dir, a, b);
    normalize3(dir);
}

static int portal_reach_tolerance(const supp_t *v1, const supp_t *v2, const supp_t *v3, const supp_t *v4,
                                  const double *dir, double tol) {
    double dv4 = dot3(v4->v, dir);
    double d1 = dv4 - dot3(v1->v, dir), d2 = dv4 - dot3(v2->v, dir), d3 = dv4 - dot3(v3->v, dir);
    double d = d1 < d2 ? d1 : d2;
    d = d < d3 ? d : d3;
    return is_zero(d) || d < tol;
}

static void expand_portal(const supp_t *v0, supp_t *v1, supp_t *v2, supp_t *v3, const supp_t *v4) {
    double v4v0[3];
    cross3(v4v0, v4->v, v0->v);
    if (dot3(v1->v, v4v0) > 0) {
        if (dot3(v2->v, v4v0) > 0) *v1 = *v4; else *v3 = *v4;
    } else {
        if (dot3(v3->v, v4v0) > 0) *v2 = *v4; else *v1 = *v4;
    }
}

/* squared distance from the origin to segment [a,b]; witness = closest point */
static double origin_segment_dist2(const double *a, const double *b, double *wit) {
    double d[3], t;
    sub3(d, b, a);
    t = -dot3(a, d);
    double dd = dot3(d, d);
    if (t <= 0 || dd < MINVAL) copy3(wit, a);
    else if (t >= dd) copy3(wit, b);
    else { copy3(wit, a); addscl3(wit, d, t / dd); }
    return dot3(wit, wit);
}

/* squared distance from the origin to triangle (a,b,c); witness = closest point */
static double origin_tri_dist2(const double *a, const double *b, const double *c, double *wit) {
    double d1[3], d2[3];
    sub3(d1, b, a);
    sub3(d2, c, a);
    double u = dot3(a, a), v = dot3(d1, d1), w = dot3(d2, d2), p = dot3(a, d1), q = dot3(a, d2), r = dot3(d1, d2);
    double den = w * v - r * r, sp = -1, tp = -1;
    if (!is_zero(den)) {
        sp = (q * r - w * p) / den;
        tp = (-sp * r - q) / w;
    }
    (void)u;
    if (sp >= 0 && sp <= 1 && tp >= 0 && tp <= 1 && sp + tp <= 1) {
        copy3(wit, a);
        addscl3(wit, d1, sp);
        addscl3(wit, d2, tp);
        return dot3(wit, wit);
    }
    double w2[3], dist = origin_segment_dist2(a, b, wit), dd;
    dd = origin_segment_dist2(a, c, w2);
    if (dd < dist) { dist = dd; copy3(wit, w2); }
    dd = origin_segment_dist2(b, c, w2);
    if (dd < dist) { dist = dd; copy3(wit, w2); }
    return dist;
}

static void find_pos(const supp_t *v0, const supp_t *v1, const supp_t *v2, const supp_t *v3, double *pos) {
    double dir[3], b[4], t[3], sum;
    portal_dir(v1, v2, v3, dir);
    cross3(t, v1->v, v2->v); b[0] = dot3(t, v3->v);
    cross3(t, v3->v, v2->v); b[1] = dot3(t, v0->v);
    cross3(t, v0->v, v1->v); b[2] = dot3(t, v3->v);
    cross3(t, v2->v, v1->v); b[3] = dot3(t, v0->v);
    sum = b[0] + b[1] + b[2] + b[3];
    if (is_zero(sum) || sum < 0) {
        b[0] = 0;
        cross3(t, v2->v, v3->v); b[1] = dot3(t, dir);
        cross3(t, v3->v, v1->v); b[2] = dot3(t, dir);
        cross3(t, v1->v, v2->v); b[3] = dot3(t, dir);
        sum = b[1] + b[2] + b[3];
    }
    double inv = 1.0 / sum, p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0};
    const supp_t *vs[4] = {v0, v1, v2, v3};
    for (int i = 0; i < 4; i++) { addscl3(p1, vs[i]->v1, b[i]); addscl3(p2, vs[i]->v2, b[i]); }
    for (int i = 0; i < 3; i++) pos[i] = 0.5 * (p1[i] + p2[i]) * inv;
}

/* Minkowski Portal Refinement (XenoCollide) penetration query between the convex hulls of
 * geoms g1,g2, each inflated by margin/2.  Returns 0 and (depth, dir g1->g2, pos) on overlap,
 * -1 otherwise.  Same structure as libccd's ccdMPRPenetration, which MuJoCo 1.50 calls for
 * mesh-mesh pairs (mpr_tolerance 1e-6, mpr_iterations 50). */
static int mpr_penetration(mpr_ctx *c, double *depth, double *dir, double *pos) {
    const ko_model *m = c->s->m;
    supp_t v0, v1, v2, v3, v4;
    double d[3], va[3], vb[3];
    c->s->mpr_calls++;
    /* ---- discover portal */
    copy3(v0.v1, c->s->geom_xpos[c->g1]);
    copy3(v0.v2, c->s->geom_xpos[c->g2]);
    sub3(v0.v, v0.v1, v0.v2);
    if (vec_is_zero(v0.v)) v0.v[0] += 1e-5;
    scl3(d, v0.v, -1.0);
    normalize3(d);
    mpr_support(c, d, &v1);
    double dt = dot3(v1.v, d);
    if (is_zero(dt) || dt < 0) return -1;
    cross3(d, v0.v, v1.v);
    if (vec_is_zero(d)) {
        if (vec_is_zero(v1.v)) return -1; /* touching contact: normal undefined -> MuJoCo drops it */
        /* origin on the v0-v1 segment */
        *depth = norm3(v1.v);
        copy3(dir, v1.v);
        normalize3(dir);
        for (int i = 0; i < 3; i++) pos[i] = 0.5 * (v1.v1[i] + v1.v2[i]);
        return 0;
    }
    normalize3(d);
    mpr_support(c, d, &v2);
    dt = dot3(v2.v, d);
    if (is_zero(dt) || dt < 0) return -1;
    sub3(va, v1.v, v0.v);
    sub3(vb, v2.v, v0.v);
    cross3(d, va, vb);
    normalize3(d);
    if (dot3(d, v0.v) > 0) { /* orient the portal away from v0 */
        supp_t t = v1; v1 = v2; v2 = t;
        scl3(d, d, -1.0);
    }
    for (int it = 0;; it++) {
        if (it > 100) return -1;
        mpr_support(c, d, &v3);
        dt = dot3(v3.v, d);
        if (is_zero(dt) || dt < 0) return -1;
        int cont = 0;
        cross3(va, v1.v, v3.v);
        dt = dot3(va, v0.v);
        if (dt < 0 && !is_zero(dt)) { v2 = v3; cont = 1; }
        if (!cont) {
            cross3(va, v3.v, v2.v);
            dt = dot3(va, v0.v);
            if (dt < 0 && !is_zero(dt)) { v1 = v3; cont = 1; }
        }
        if (!cont) break;
        sub3(va, v1.v, v0.v);
        sub3(vb, v2.v, v0.v);
        cross3(d, va, vb);
        normalize3(d);
    }
    /* ---- refine portal until it encloses the origin */
    for (int it = 0;; it++) {
        if (it > 100) return -1;
        portal_dir(&v1, &v2, &v3, d);
        dt = dot3(d, v1.v);
        if (is_zero(dt) || dt > 0) break; /* portal encapsulates origin */
        mpr_support(c, d, &v4);
        dt = dot3(v4.v, d);
        if (!(is_zero(dt) || dt > 0)) return -1; /* cannot reach the origin */
        if (portal_reach_tolerance(&v1, &v2, &v3, &v4, d, m->mpr_tolerance)) return -1;
        expand_portal(&v0, &v1, &v2, &v3, &v4);
    }
    /* ---- penetration: push the portal to the surface */
    for (int it = 0;; it++) {
        portal_dir(&v1, &v2, &v3, d);
        mpr_support(c, d, &v4);
        if (portal_reach_tolerance(&v1, &v2, &v3, &v4, d, m->mpr_tolerance) || it > m->mpr_iterations) {
            double wit[3];
            *depth = sqrt(origin_tri_dist2(v1.v, v2.v, v3.v, wit));
            if (vec_is_zero(wit)) return -1; /* normal undefined */
            copy3(dir, wit);
            normalize3(dir);
            find_pos(&v0, &v1, &v2, &v3, pos);
            return 0;
        }
        expand_portal(&v0, &v1, &v2, &v3, &v4);
    }
}


/* ---- GJK closest-features query on the UN-inflated hulls.
 * Contacts live in the margin zone (0 <= dist < margin) almost all the time; there the closest
 * points of two convex polytopes are unique and well conditioned, whereas MPR on margin-inflated
 * supports mixes inflation directions and returns normals that are sensitive to round-off.  MuJoCo
 * >= 3.2 ("native ccd") makes the same choice; MuJoCo 1.50 used libccd MPR for both regimes.
 * Returns 0: separated by >= margin (no contact), 1: contact with dist in (0, margin),
 * 2: hulls overlap (caller falls back to MPR without inflation). */
typedef struct { double y[4][3], a[4][3], b[4][3]; int n; } gjk_simplex;

static void gjk_support(mpr_ctx *c, const double *dir, double *y, double *a, double *b) {
    double nd[3] = {-dir[0], -dir[1], -dir[2]};
    hull_support(c->s, c->g1, dir, 0.0, a);
    hull_support(c->s, c->g2, nd, 0.0, b);
    sub3(y, a, b);
    c->s->mpr_support_calls++;
}

/* closest point to the origin on the segment A + t (B - A), t clamped to [0, 1]; returns its squared norm */
static double closest_seg(const double *A, const double *B, double *t) {
    double d[3], q[3];
    sub3(d, B, A);
    double dd = dot3(d, d), tt = -dot3(A, d);
    *t = dd > 1e-30 ? (tt <= 0 ? 0.0 : (tt >= dd ? 1.0 : tt / dd)) : 0.0;
    copy3(q, A);
    addscl3(q, d, *t);
    return dot3(q, q);
}

/* closest point to the origin on triangle (A,B,C): barycentric l[3].  Minimum over four candidates that are all
 * points of the triangle: the closest points of the three clamped edges and, when the origin projects inside, the
 * interior point.  (The textbook region tests - Ericson, RTCD 5.1.5 - decide on products such as d1 d4 - d3 d2 that
 * cancel for the sliver triangles of a Minkowski difference; in single precision a wrong region ends GJK early with a
 * normal several degrees off.  The kernels use this same formulation so that both sides break ties alike.) */
static void closest_tri(const double *A, const double *B, const double *C, double *l) {
    double tab, tac, tbc;
    double dab = closest_seg(A, B, &tab), dac = closest_seg(A, C, &tac), dbc = closest_seg(B, C, &tbc);
    double best = dab;
    l[0] = 1 - tab; l[1] = tab; l[2] = 0;
    if (dac < best) { best = dac; l[0] = 1 - tac; l[1] = 0; l[2] = tac; }
    if (dbc < best) { best = dbc; l[0] = 0; l[1] = 1 - tbc; l[2] = tbc; }
    double ab[3], ac[3], n[3], c1[3], c2[3];
    sub3(ab, B, A); sub3(ac, C, A);
    cross3(n, ab, ac);
    double nn = dot3(n, n);
    if (nn > 1e-30) {
        cross3(c1, ac, A);
        cross3(c2, A, ab);
        double l1 = dot3(c1, n) / nn, l2 = dot3(c2, n) / nn, l0 = 1 - l1 - l2;
        if (l0 > 0 && l1 > 0 && l2 > 0) {
            double q[3] = {0, 0, 0};
            addscl3(q, A, l0); addscl3(q, B, l1); addscl3(q, C, l2);
            if (dot3(q, q) < best) { l[0] = l0; l[1] = l1; l[2] = l2; }
        }
    }
}

/* closest point to the origin on the simplex; reduces it to the supporting sub-simplex, writes
 * the barycentric weights of the kept vertices.  Returns 1 if the origin is inside a tetrahedron. */
static int gjk_closest(gjk_simplex *S, double *lam, double *v) {
    double l[4] = {0, 0, 0, 0};
    int keep[4] = {0, 1, 2, 3}, nk = S->n;
    if (S->n == 1) { l[0] = 1; }
    else if (S->n == 2) {
        double d[3];
        sub3(d, S->y[1], S->y[0]);
        double t = -dot3(S->y[0], d), dd = dot3(d, d);
        if (t <= 0 || dd < MINVAL) { l[0] = 1; l[1] = 0; }
        else if (t >= dd) { l[0] = 0; l[1] = 1; }
        else { l[1] = t / dd; l[0] = 1 - l[1]; }
    } else if (S->n == 3) closest_tri(S->y[0], S->y[1], S->y[2], l);
    else {
        /* tetrahedron: test the faces the origin lies outside of, keep the nearest */
        static const int F[4][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 3, 2, 0}};
        double best = 1e300;
        int any = 0;
        for (int f = 0; f < 4; f++) {
            const double *A = S->y[F[f][0]], *B = S->y[F[f][1]], *C = S->y[F[f][2]], *D = S->y[F[f][3]];
            double ab[3], ac[3], n[3], ad[3];
            sub3(ab, B, A); sub3(ac, C, A); cross3(n, ab, ac); sub3(ad, D, A);
            double sp = -dot3(A, n), sd = dot3(ad, n);
            /* a sliver (4th vertex within 1e-3 rad of the face plane) cannot certify "inside": its
             * faces are all evaluated instead, which keeps fp32 and fp64 on the same branch */
            int flat = sd * sd <= 1e-6 * dot3(n, n) * dot3(ad, ad);
            if (sp * sd < 0 || flat) { /* origin outside this face (or sliver tetra) */
                double lt[3], q[3] = {0, 0, 0};
                closest_tri(A, B, C, lt);
                addscl3(q, A, lt[0]); addscl3(q, B, lt[1]); addscl3(q, C, lt[2]);
                double d2 = dot3(q, q);
                if (d2 < best) {
                    best = d2; any = 1;
                    l[0] = l[1] = l[2] = l[3] = 0;
                    l[F[f][0]] = lt[0]; l[F[f][1]] = lt[1]; l[F[f][2]] = lt[2];
                }
            }
        }
        if (!any) return 1;
    }
    /* compact */
    gjk_simplex R;
    R.n = 0;
    for (int i = 0; i < nk; i++)
        if (l[keep[i]] > 0) {
            copy3(R.y[R.n], S->y[i]); copy3(R.a[R.n], S->a[i]); copy3(R.b[R.n], S->b[i]);
            lam[R.n] = l[i];
            R.n++;
        }
    *S = R;
    v[0] = v[1] = v[2] = 0;
    for (int i = 0; i < S->n; i++) addscl3(v, S->y[i], lam[i]);
    return 0;
}

static int gjk_distance(mpr_ctx *c, double margin, double *dist, double *normal, double *pos) {
    gjk_simplex S;
    double lam[4] = {1, 0, 0, 0}, v[3], d[3];
    const double tol = KO_GJK_TOL; /* relative progress tolerance; polytopes normally stop on a repeated vertex */
    sub3(d, c->s->geom_xpos[c->g2], c->s->geom_xpos[c->g1]);
    if (dot3(d, d) < MINVAL) { d[0] = 1; d[1] = 0; d[2] = 0; }
    gjk_support(c, d, S.y[0], S.a[0], S.b[0]);
    S.n = 1;
    copy3(v, S.y[0]);
    double last_vw = 1.0;
    for (int it = 0; it < 48; it++) {
        double vv = dot3(v, v);
        if (vv < 1e-24) return 2;
        double nd[3] = {-v[0], -v[1], -v[2]}, w[3], wa[3], wb[3];
        gjk_support(c, nd, w, wa, wb);
        double vw = dot3(v, w);
        last_vw = vw;
        if (vw > 0 && vw * vw >= margin * margin * vv) return 0; /* separating plane beyond the margin */
        if (vv - vw <= tol * vv) break;                          /* no further progress possible */
        int dup = 0;
        for (int i = 0; i < S.n; i++)
            if (S.y[i][0] == w[0] && S.y[i][1] == w[1] && S.y[i][2] == w[2]) dup = 1;
        if (dup) break;
        copy3(S.y[S.n], w); copy3(S.a[S.n], wa); copy3(S.b[S.n], wb);
        S.n++;
        gjk_simplex prev = S;
        double plam[4] = {lam[0], lam[1], lam[2], lam[3]}, pv[3] = {v[0], v[1], v[2]};
        (void)prev;
        if (gjk_closest(&S, lam, v)) return 2;
        if (dot3(v, v) >= vv) { /* round-off: no decrease -> keep the previous iterate */
            copy3(v, pv);
            S = prev; S.n--;
            lam[0] = plam[0]; lam[1] = plam[1]; lam[2] = plam[2]; lam[3] = plam[3];
            break;
        }
    }
    /* the last support point lay beyond the origin along -v: v is no certified separation (a flat tetrahedron of a near-touching
     * pair can end the iteration a few um "apart" while the hulls overlap) - the penetration query decides first (return 3 /
     * 2), the caller keeps this result only if that finds no overlap.  Same rule in the kernels (ks_core.h gjk_distance). */
    int open = last_vw < 0;
    double dd = norm3(v);
    if (dd < 1e-12) return 2;
    if (dd >= margin) return open ? 2 : 0;
    double p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0};
    for (int i = 0; i < S.n; i++) { addscl3(p1, S.a[i], lam[i]); addscl3(p2, S.b[i], lam[i]); }
    *dist = dd;
    for (int i = 0; i < 3; i++) { normal[i] = -v[i] / dd; pos[i] = 0.5 * (p1[i] + p2[i]); }
    return open ? 3 : 1;
}

/* MuJoCo's mju_makeFrame: complete a contact frame from its normal */
static void make_frame(double *f) {
    normalize3(f);
    double *y = f + 3, *z = f + 6;
    y[0] = y[1] = y[2] = 0;
    if (f[1] < 0.5 && f[1] > -0.5) y[1] = 1; else y[2] = 1;
    double t = dot3(f, y);
    addscl3(y, f, -t);
    normalize3(y);
    cross3(z, f, y);
}

static void add_contact(ko_sim *s, int g1, int g2, const double *pair, double dist, const double *pos, const double *normal) {
    if (s->ncon >= s->ncon_max) { s->ncon_dropped++; return; }
    ko_contact *c = &s->contact[s->ncon++];
    c->dist = dist; c->geom1 = g1; c->geom2 = g2;
    c->mu[0] = pair[2]; c->mu[1] = pair[3]; c->margin = pair[4];
    if (s->obj_mu > 0 && g1 != 0 && g2 >= KO_OBJ_GEOM) c->mu[0] = c->mu[1] = s->obj_mu; /* object-hand pairs */
    copy3(c->pos, pos);
    copy3(c->frame, normal);
    make_frame(c->frame);
}

/* plane (geom 0, normal +z through its origin) vs convex hull: deepest vertex first, then up to
 * three more vertices within the margin that are > 0.3*rbound away from every accepted one. */
static void collide_plane_hull(ko_sim *s, int g2, const double *pair) {
    const ko_model *m = s->m;
    const double *n = &s->geom_xmat[0][0]; /* plane normal = z column of the plane frame */
    double normal[3] = {n[2], n[5], n[8]}, margin = pair[4];
    double cdist, t[3];
    sub3(t, s->geom_xpos[g2], s->geom_xpos[0]);
    cdist = dot3(t, normal);
    if (cdist > m->geom_rbound[g2] + margin) return;
    int mesh = m->geom_mesh[g2], nv = m->mesh_nvert[mesh];
    const double *V = m->mesh_vert[mesh];
    double ln[3];
    mulmatTvec3(ln, s->geom_xmat[g2], normal);
    /* vertex distance = cdist + v.ln */
    int best = 0;
    double bd = 1e300;
    for (int i = 0; i < nv; i++) {
        double d = cdist + V[3 * i] * ln[0] + V[3 * i + 1] * ln[1] + V[3 * i + 2] * ln[2];
        if (d < bd) { bd = d; best = i; }
    }
    if (bd > margin) return;
    /* Ties: a cylinder standing on its base has 64 rim vertices at the same depth to the last bits, a landing cube four - "the
     * deepest" is then decided by rounding, differently in every arithmetic, and with it the greedy choice of the other three
     * contacts.  Rule (shared with the fp64 instantiation of the kernels): the first contact is the LOWEST-INDEX vertex within
     * PLANE_TIE_EPS of the deepest one.  1e-12 m is four orders above fp64 rounding of these distances and has no physical effect
     * (the 1 um dead band tried in round 3 kept round objects rocking: tools/experiments/r03_plane_tie_rule.patch). */
    for (int i = 0; i < nv; i++) {
        double d = cdist + V[3 * i] * ln[0] + V[3 * i + 1] * ln[1] + V[3 * i + 2] * ln[2];
        if (d <= bd + PLANE_TIE_EPS) { best = i; break; }
    }
    int chosen[4], nc = 0;
    chosen[nc++] = best;
    double thr2 = PLANE_MESH_TOL * m->geom_rbound[g2];
    thr2 *= thr2;
    for (int i = 0; i < nv && nc < 4; i++) {
        double d = cdist + V[3 * i] * ln[0] + V[3 * i + 1] * ln[1] + V[3 * i + 2] * ln[2];
        if (d > margin) continue;
        int ok = 1;
        for (int k = 0; k < nc; k++) {
            double dv[3];
            sub3(dv, &V[3 * i], &V[3 * chosen[k]]);
            if (dot3(dv, dv) <= thr2) { ok = 0; break; }
        }
        if (ok) chosen[nc++] = i;
    }
    for (int k = 0; k < nc; k++) {
        const double *v = &V[3 * chosen[k]];
        double d = cdist + v[0] * ln[0] + v[1] * ln[1] + v[2] * ln[2], w[3], pos[3];
        mulmatvec3(w, s->geom_xmat[g2], v);
        add3(w, w, s->geom_xpos[g2]);
        copy3(pos, w);
        addscl3(pos, normal, -0.5 * d);
        add_contact(s, 0, g2, pair, d, pos, normal);
    }
}

static void collide_hull_hull(ko_sim *s, int g1, int g2, const double *pair) {
    const ko_model *m = s->m;
    double margin = pair[4], t[3];
    sub3(t, s->geom_xpos[g1], s->geom_xpos[g2]);
    double bound = m->geom_rbound[g1] + m->geom_rbound[g2] + margin;
    if (dot3(t, t) > bound * bound) return;
    if (s->narrow_phase == 1) {
        /* study mode (tests/studies/narrow_phase.py): MuJoCo 1.50's own scheme - libccd MPR on hulls inflated by margin / 2 each, in
         * the margin zone and in overlap alike; distance = margin - (penetration of the inflated hulls).  NOT what the product does
         * (DESIGN.md section 2): kept to quantify that deviation. */
        const int of = (g2 == KO_OBJ_GEOM); /* operand order: see below */
        mpr_ctx ci = {s, of ? g2 : g1, of ? g1 : g2, 0.5 * margin};
        double depth_i, dir_i[3], pos_i[3];
        if (mpr_penetration(&ci, &depth_i, dir_i, pos_i) == 0) {
            if (of) scl3(dir_i, dir_i, -1.0);
            add_contact(s, g1, g2, pair, margin - depth_i, pos_i, dir_i);
        }
        return;
    }
    /* Operand order of the convex queries = MuJoCo's.  mj_collideGeoms hands the narrow phase the geoms of an explicit <pair> in
     * the XML's order (geom1 = "object", geom2 = the hand geom: XML:159-166) unless geom1's TYPE is the larger one (never here:
     * mesh 7 / box 6 / cylinder 5 object against a mesh): the OBJECT is libccd's obj1.  The model stores every pair as
     * (lower geom id, higher geom id), i.e. (hand geom, object): for those pairs the queries run with the operands exchanged and
     * the direction is flipped back to the stored pair's g1 -> g2.  MPR is not symmetric in its operands (portal discovery and
     * expansion use cross products, which a point reflection does not mirror): within its 1e-6 tolerance the path - depth,
     * normal, contact point of a pad-on-face contact - depends on the order.  Pinned by the recorded MuJoCo 1.50 trajectory
     * (tests/test_mujoco_recorded.py): rows 35-45 (two and three finger pads on the box's faces, grasp and lift) agree to 2e-10 /
     * 8e-8 with the object first and only to 1.3e-6 with the hand geom first; edge contacts (rows 4-34) do not depend on it.
     * Dynamically generated pairs (hand vs hand, hand vs welded object pieces) reach the narrow phase in body / geom order:
     * lower id first, as stored. */
    const int obj_first = (g2 == KO_OBJ_GEOM);
    mpr_ctx c = {s, obj_first ? g2 : g1, obj_first ? g1 : g2, 0.0};
    double depth, dist = 0, dir[3] = {0, 0, 0}, pos[3] = {0, 0, 0};
    int r = gjk_distance(&c, margin, &dist, dir, pos);
    if (obj_first) scl3(dir, dir, -1.0);
    if (r == 1) add_contact(s, g1, g2, pair, dist, pos, dir);
    else if (r >= 2) {
        double mdir[3], mpos[3];
        if (mpr_penetration(&c, &depth, mdir, mpos) == 0) {
            if (obj_first) scl3(mdir, mdir, -1.0);
            add_contact(s, g1, g2, pair, -depth, mpos, mdir);
        } else if (r == 3) add_contact(s, g1, g2, pair, dist, pos, dir);
    }
}

static void collision(ko_sim *s) {
    const ko_model *m = s->m;
    s->ncon = 0;
    s->ncon_dropped = 0;
    for (int p = 0; p < m->npair; p++) {
        int g1 = (int)m->pairs[p][0], g2 = (int)m->pairs[p][1];
        if (g1 == 0) collide_plane_hull(s, g2, m->pairs[p]);
        else collide_hull_hull(s, g1, g2, m->pairs[p]);
    }
}

/* ------------------------------------------------------------------ S5 constraints */
static double impedance(const double *solimp, double x) {
    /* d(r): dmin at 0 rising smoothly to dmax at |r| = width (midpoint 0.5, power 2 sigmoid) */
    double dmin = solimp[0], dmax = solimp[1], width = solimp[2], y;
    x = fabs(x) / width;
    if (x >= 1) return dmax;
    if (x <= 0.5) y = 2 * x * x; else y = 1 - 2 * (1 - x) * (1 - x);
    return dmin + y * (dmax - dmin);
}

static void add_row(ko_sim *s, int type, const double *J, double pos, double margin, double diag_approx) {
    const ko_model *m = s->m;
    int i = s->nefc++;
    s->efc_type[i] = type;
    memcpy(s->efc_J[i], J, KO_NV * sizeof(double));
    s->efc_pos[i] = pos;
    s->efc_margin[i] = margin;
    /* reference acceleration  aref = -b*(J qvel) - k*d*(pos - margin)   (timeconst >= 2*dt: refsafe) */
    double tc = m->solref[0] < 2 * m->dt ? 2 * m->dt : m->solref[0], dr = m->solref[1], dmax = m->solimp[1];
    double k = 1.0 / (dmax * dmax * tc * tc * dr * dr), b = 2.0 / (dmax * tc);
    double imp = impedance(m->solimp, pos - margin), vel = 0;
    for (int j = 0; j < KO_NV; j++) vel += J[j] * s->qvel[j];
    s->efc_aref[i] = -b * vel - k * imp * (pos - margin);
    double R = (1 - imp) / imp * diag_approx;
    s->efc_R[i] = R > MINVAL ? R : MINVAL;
}

static void make_constraint(ko_sim *s) {
    const ko_model *m = s->m;
    double J[KO_NV];
    s->nefc = 0;
    /* tendon equalities: L = c0*q_prox + c1*q_dist held at its qpos0 length 0 (XML:171-188) */
    for (int t = 0; t < 3; t++) {
        memset(J, 0, sizeof J);
        J[3 + 2 * t] = m->tendon_coef[t][0];
        J[4 + 2 * t] = m->tendon_coef[t][1];
        double len = m->tendon_coef[t][0] * s->qpos[3 + 2 * t] + m->tendon_coef[t][1] * s->qpos[4 + 2 * t];
        add_row(s, 0, J, len, 0.0, m->tendon_invweight0[t]);
    }
    /* joint limits (margin 0): slides, then hinges flagged limited */
    for (int j = 0; j < 9; j++) {
        double lo, hi;
        if (j < 3) { lo = m->slide_range[j][0]; hi = m->slide_range[j][1]; }
        else { if (!m->hinge_limited[j - 3]) continue; lo = m->hinge_range[j - 3][0]; hi = m->hinge_range[j - 3][1]; }
        double q = s->qpos[j];
        memset(J, 0, sizeof J);
        if (q - lo < 0) { J[j] = 1; add_row(s, 1, J, q - lo, 0.0, m->dof_invweight0[j]); }
        if (hi - q < 0) { J[j] = -1; add_row(s, 1, J, hi - q, 0.0, m->dof_invweight0[j]); }
    }
    /* contacts: pyramidal cone, condim 3 -> 4 rows  Jn +- mu1*Jt1, Jn +- mu2*Jt2 */
    for (int ci = 0; ci < s->ncon; ci++) {
        const ko_contact *c = &s->contact[ci];
        if (c->dist >= c->margin) continue;
        int b1 = m->geom_body[c->geom1], b2 = m->geom_body[c->geom2];
        double Jp1[3][KO_NV], Jp2[3][KO_NV], Jr[3][KO_NV], Jd[3][KO_NV];
        jac_point(s, b1, c->pos, Jp1, Jr);
        jac_point(s, b2, c->pos, Jp2, Jr);
        for (int a = 0; a < 3; a++) /* rows of the contact frame applied to (v2 - v1) */
            for (int j = 0; j < KO_NV; j++)
                Jd[a][j] = c->frame[3 * a] * (Jp2[0][j] - Jp1[0][j]) + c->frame[3 * a + 1] * (Jp2[1][j] - Jp1[1][j]) +
                           c->frame[3 * a + 2] * (Jp2[2][j] - Jp1[2][j]);
        double w = geom_invweight_of(s, c->geom1) + geom_invweight_of(s, c->geom2), mu = c->mu[0];
        double diag = (w + mu * mu * w) * 2 * mu * mu / m->impratio;
        for (int k = 0; k < 4; k++) {
            double sgn = (k & 1) ? -1.0 : 1.0, muk = c->mu[k >> 1];
            const double *Jt = Jd[1 + (k >> 1)];
            for (int j = 0; j < KO_NV; j++) J[j] = Jd[0][j] + sgn * muk * Jt[j];
            add_row(s, 2, J, c->dist, c->margin, diag);
        }
    }
}

/* ------------------------------------------------------------------ S6 solver: PGS (comparison only) */
static void solve_pgs(ko_sim *s) {
    int n = s->nefc;
    double A[KO_NEFC_MAX][KO_NEFC_MAX], B[KO_NEFC_MAX][KO_NV]; /* ~110 KB of stack, thread-safe */
    double *f = s->efc_force;
    /* B = (M^-1 J^T)^T rows; A = J M^-1 J^T + diag(R); b = J qacc_smooth - aref */
    for (int i = 0; i < n; i++) chol_solve(s->L, s->efc_J[i], B[i]);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) {
            double a = 0;
            for (int k = 0; k < KO_NV; k++) a += s->efc_J[i][k] * B[j][k];
            A[i][j] = a;
        }
        A[i][i] += s->efc_R[i];
        double b = -s->efc_aref[i];
        for (int k = 0; k < KO_NV; k++) b += s->efc_J[i][k] * s->qacc_smooth[k];
        s->efc_b[i] = b;
    }
    /* warm start: forces implied by qacc_warmstart through the primal map, kept only if they
     * beat the cold start in dual cost */
    double cost = 0;
    for (int i = 0; i < n; i++) {
        double jar = -s->efc_aref[i];
        for (int k = 0; k < KO_NV; k++) jar += s->efc_J[i][k] * s->qacc_warmstart[k];
        double fi = -jar / s->efc_R[i];
        if (s->efc_type[i] != 0 && fi < 0) fi = 0;
        f[i] = fi;
    }
    for (int i = 0; i < n; i++) {
        double Af = 0;
        for (int j = 0; j < n; j++) Af += A[i][j] * f[j];
        cost += f[i] * (0.5 * Af + s->efc_b[i]);
    }
    if (cost > 0) for (int i = 0; i < n; i++) f[i] = 0;
    /* projected Gauss-Seidel sweeps, row order = efc order */
    for (int it = 0; it < s->solver_iterations; it++)
        for (int i = 0; i < n; i++) {
            double res = s->efc_b[i];
            for (int j = 0; j < n; j++) res += A[i][j] * f[j];
            double fi = f[i] - res / A[i][i];
            if (s->efc_type[i] != 0 && fi < 0) fi = 0;
            f[i] = fi;
        }
    memset(s->qfrc_constraint, 0, sizeof s->qfrc_constraint);
    for (int k = 0; k < KO_NV; k++) s->qacc[k] = s->qacc_smooth[k];
    for (int i = 0; i < n; i++)
        for (int k = 0; k < KO_NV; k++) {
            s->qfrc_constraint[k] += s->efc_J[i][k] * f[i];
            s->qacc[k] += B[i][k] * f[i];
        }
}


/* ------------------------------------------------------------------ S6 solver: Newton (default)
 * MuJoCo's default solver for this model (the XML sets none): minimise over qacc
 *     1/2 (a - a_s)^T M (a - a_s) + sum_i s_i(J_i a - aref_i),   s_i(x) = 1/2 D_i x^2 on active rows
 * (equality rows always active, limit / pyramid rows active when x < 0; D_i = 1/R_i), with an
 * exact line search on the piecewise-quadratic 1-D restriction.  The minimiser is unique, so
 * fp32 and fp64 implementations agree to round-off once converged.  A fixed iteration count
 * (no early exit) keeps the GPU lanes in lock step and the result deterministic. */
static void chol_n(int n, const double *A, double *L) {
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k];
        L[j * n + j] = sqrt(d > MINVAL ? d : MINVAL);
        for (int i = j + 1; i < n; i++) {
            double v = A[i * n + j];
            for (int k = 0; k < j; k++) v -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = v / L[j * n + j];
        }
    }
}

static double primal_cost(const ko_sim *s, const double *a) {
    double c = 0, d[KO_NV];
    for (int i = 0; i < KO_NV; i++) d[i] = a[i] - s->qacc_smooth[i];
    for (int i = 0; i < KO_NV; i++)
        for (int j = 0; j < KO_NV; j++) c += 0.5 * d[i] * s->M[i][j] * d[j];
    for (int i = 0; i < s->nefc; i++) {
        double jar = -s->efc_aref[i];
        for (int k = 0; k < KO_NV; k++) jar += s->efc_J[i][k] * a[k];
        if (s->efc_type[i] == 0 || jar < 0) c += 0.5 * jar * jar / s->efc_R[i];
    }
    return c;
}

static void solve_newton(ko_sim *s) {
    int n = s->nefc;
    double a[KO_NV], jar[KO_NEFC_MAX], jp[KO_NEFC_MAX], g[KO_NV], p[KO_NV], Ma[KO_NV], Mp[KO_NV];
    double H[KO_NV * KO_NV], Lh[KO_NV * KO_NV], qfrc_smooth[KO_NV];
    for (int i = 0; i < KO_NV; i++) {
        qfrc_smooth[i] = 0;
        for (int j = 0; j < KO_NV; j++) qfrc_smooth[i] += s->M[i][j] * s->qacc_smooth[j];
    }
    /* warm start: previous qacc unless the unconstrained acceleration is cheaper */
    if (primal_cost(s, s->qacc_warmstart) < primal_cost(s, s->qacc_smooth)) memcpy(a, s->qacc_warmstart, sizeof a);
    else memcpy(a, s->qacc_smooth, sizeof a);
    s->newton_last_grad = 0;
    s->newton_converged = 0;
    s->newton_iters_used = 0;
    for (int it = 0; it < s->solver_iterations; it++) {
        /* gradient and Hessian at a */
        for (int i = 0; i < KO_NV; i++) {
            Ma[i] = -qfrc_smooth[i];
            for (int j = 0; j < KO_NV; j++) Ma[i] += s->M[i][j] * a[j];
            g[i] = Ma[i];
            for (int j = 0; j < KO_NV; j++) H[i * KO_NV + j] = s->M[i][j];
        }
        for (int i = 0; i < n; i++) {
            double x = -s->efc_aref[i];
            for (int k = 0; k < KO_NV; k++) x += s->efc_J[i][k] * a[k];
            jar[i] = x;
            if (s->efc_type[i] == 0 || x < 0) {
                double D = 1.0 / s->efc_R[i];
                for (int k = 0; k < KO_NV; k++) {
                    g[k] += s->efc_J[i][k] * D * x;
                    for (int l = 0; l < KO_NV; l++) H[k * KO_NV + l] += D * s->efc_J[i][k] * s->efc_J[i][l];
                }
            }
        }
        double gn = 0;
        for (int k = 0; k < KO_NV; k++) gn += g[k] * g[k];
        s->newton_last_grad = sqrt(gn);
        /* Newton direction p = -H^-1 g */
        chol_n(KO_NV, H, Lh);
        {
            double y[KO_NV];
            for (int i = 0; i < KO_NV; i++) {
                double v = -g[i];
                for (int k = 0; k < i; k++) v -= Lh[i * KO_NV + k] * y[k];
                y[i] = v / Lh[i * KO_NV + i];
            }
            for (int i = KO_NV - 1; i >= 0; i--) {
                double v = y[i];
                for (int k = i + 1; k < KO_NV; k++) v -= Lh[k * KO_NV + i] * p[k];
                p[i] = v / Lh[i * KO_NV + i];
            }
        }
        /* exact line search: root of phi'(alpha), phi' piecewise linear and increasing */
        double pMa = 0, pMp = 0;
        for (int i = 0; i < KO_NV; i++) {
            Mp[i] = 0;
            for (int j = 0; j < KO_NV; j++) Mp[i] += s->M[i][j] * p[j];
            pMa += p[i] * Ma[i];
            pMp += p[i] * Mp[i];
        }
        for (int i = 0; i < n; i++) {
            double x = 0;
            for (int k = 0; k < KO_NV; k++) x += s->efc_J[i][k] * p[k];
            jp[i] = x;
        }
        double alpha = 0, lo = 0, hi = -1; /* hi < 0: no upper bracket yet */
        for (int ls = 0; ls < 30; ls++) {
            double d1 = pMa + alpha * pMp, d2 = pMp;
            for (int i = 0; i < n; i++) {
                double x = jar[i] + alpha * jp[i];
                if (s->efc_type[i] == 0 || x < 0) {
                    double D = 1.0 / s->efc_R[i];
                    d1 += D * x * jp[i];
                    d2 += D * jp[i] * jp[i];
                }
            }
            if (d2 < MINVAL) break;
            if (d1 < 0) lo = alpha; else hi = alpha;
            double next = alpha - d1 / d2;
            if (hi >= 0 && (next < lo || next > hi)) next = 0.5 * (lo + hi); /* safeguard: bisect */
            if (next < lo) next = lo;
            if (fabs(next - alpha) <= 1e-14 * (1 + fabs(alpha))) { alpha = next; break; }
            alpha = next;
        }
        double amax = 0, dmax = 0;
        for (int i = 0; i < KO_NV; i++) {
            double da = alpha * p[i];
            a[i] += da;
            if (fabs(a[i]) > amax) amax = fabs(a[i]);
            if (fabs(da) > dmax) dmax = fabs(da);
        }
        s->newton_iters_used = it + 1;
        s->newton_last_step = dmax / (1 + amax);
        if (dmax <= s->solver_tolerance * (1 + amax)) { s->newton_converged = 1; break; } /* same rule as the GPU kernel */
    }
    /* outputs */
    memcpy(s->qacc, a, sizeof a);
    memset(s->qfrc_constraint, 0, sizeof s->qfrc_constraint);
    for (int i = 0; i < n; i++) {
        double x = -s->efc_aref[i];
        for (int k = 0; k < KO_NV; k++) x += s->efc_J[i][k] * a[k];
        double f = (s->efc_type[i] == 0 || x < 0) ? -x / s->efc_R[i] : 0.0;
        s->efc_force[i] = f;
        for (int k = 0; k < KO_NV; k++) s->qfrc_constraint[k] += s->efc_J[i][k] * f;
    }
}

/* ------------------------------------------------------------------ S8 sensors */
/* Ray vs mesh geom as MuJoCo's mj_rayMesh does it: bounding-box pre-test (geom_size about the geom
 * origin), then EVERY face of the original triangle mesh (not the convex hull), both orientations;
 * nearest intersection with t >= 0, -1 if none. */
static double ray_mesh(const ko_sim *s, int g, const double *pnt, const double *vec) {
    const ko_model *m = s->m;
    int mesh = m->geom_mesh[g], nt = m->mesh_ntri[mesh];
    const double *T = m->mesh_tri[mesh], *e = m->geom_size[g];
    double lp[3], lv[3], t[3];
    sub3(t, pnt, s->geom_xpos[g]);
    mulmatTvec3(lp, s->geom_xmat[g], t);
    mulmatTvec3(lv, s->geom_xmat[g], vec);
    double t0 = 0, t1 = 1e300; /* slab test against the box */
    for (int a = 0; a < 3; a++) {
        if (fabs(lv[a]) < MINVAL) { if (fabs(lp[a]) > e[a]) return -1; continue; }
        double ta = (-e[a] - lp[a]) / lv[a], tb = (e[a] - lp[a]) / lv[a];
        if (ta > tb) { double w = ta; ta = tb; tb = w; }
        if (ta > t0) t0 = ta;
        if (tb < t1) t1 = tb;
        if (t0 > t1) return -1;
    }
    double best = -1;
    for (int i = 0; i < nt; i++) {
        const double *v0 = &T[9 * i], *v1 = v0 + 3, *v2 = v0 + 6;
        double e1[3], e2[3], pv[3], tv[3], qv[3];
        sub3(e1, v1, v0); sub3(e2, v2, v0);
        cross3(pv, lv, e2);
        double det = dot3(e1, pv);
        if (fabs(det) < 1e-30) continue;
        double inv = 1.0 / det;
        sub3(tv, lp, v0);
        double u = dot3(tv, pv) * inv;
        if (u < 0 || u > 1) continue;
        cross3(qv, tv, e1);
        double v = dot3(lv, qv) * inv;
        if (v < 0 || u + v > 1) continue;
        double tt = dot3(e2, qv) * inv;
        if (tt >= 0 && (best < 0 || tt < best)) best = tt;
    }
    return best;
}

static void sensors(ko_sim *s) {
    const ko_model *m = s->m;
    /* jointpos x9 in sensor order (XML:247-262): slides, proximal 1-3, distal 1-3 */
    for (int k = 0; k < 3; k++) {
        s->sensordata[k] = s->qpos[k];
        s->sensordata[3 + k] = s->qpos[3 + 2 * k];
        s->sensordata[6 + k] = s->qpos[4 + 2 * k];
    }
    /* rangefinders (XML:265-288): ray along site +z, geoms of the site's own body excluded */
    if (!s->rays_enabled) return;
    for (int i = 0; i < KO_NSITE; i++) {
        const double *pnt = s->site_xpos[i];
        double vec[3] = {s->site_xmat[i][2], s->site_xmat[i][5], s->site_xmat[i][8]}, best = -1;
        for (int g = 0; g < m->ngeom; g++) {
            if (m->geom_body[g] == m->site_body[i]) continue;
            double d = -1;
            if (g == 0) { /* ground plane z=0, finite half-size (XML:148).  mju_rayGeom's plane case: a ray whose local z component
                           * does not point at the FRONT face (+z side) is rejected - a site below the floor looking up (the
                           * 'rotated' / 'top' fresh-env starts) sees no ground */
                if (vec[2] < -MINVAL) {
                    double t = -(pnt[2] - s->geom_xpos[0][2]) / vec[2];
                    if (t >= 0) {
                        double x = pnt[0] + t * vec[0], y = pnt[1] + t * vec[1];
                        if (fabs(x) <= m->geom_size[0][0] && fabs(y) <= m->geom_size[0][1]) d = t;
                    }
                }
            } else d = ray_mesh(s, g, pnt, vec);
            if (d >= 0 && (best < 0 || d < best)) best = d;
        }
        s->sensordata[9 + i] = best;
    }
}

/* ------------------------------------------------------------------ forward / step */
void ko_forward(ko_sim *s) {
    ko_kinematics(s);
    mass_matrix(s);
    cholesky(s->M, s->L);
    collision(s);
    make_constraint(s);
    sensors(s);
    bias_forces(s);
    passive_and_actuation(s);
    double f[KO_NV];
    for (int i = 0; i < KO_NV; i++) f[i] = s->qfrc_passive[i] - s->qfrc_bias[i] + s->qfrc_actuator[i];
    chol_solve(s->L, f, s->qacc_smooth);
    if (s->solver == 1) solve_pgs(s); else solve_newton(s);
}

/* S7: semi-implicit Euler with joint damping treated implicitly */
static void euler(ko_sim *s) {
    const ko_model *m = s->m;
    double h = m->dt, Md[KO_NV][KO_NV], Ld[KO_NV][KO_NV], f[KO_NV], qacc[KO_NV];
    memcpy(s->qacc_warmstart, s->qacc, sizeof s->qacc);
    memcpy(Md, s->M, sizeof Md);
    for (int i = 0; i < KO_NV; i++) {
        Md[i][i] += h * m->dof_damping[i];
        f[i] = s->qfrc_passive[i] - s->qfrc_bias[i] + s->qfrc_actuator[i] + s->qfrc_constraint[i];
    }
    cholesky(Md, Ld);
    chol_solve(Ld, f, qacc);
    for (int i = 0; i < KO_NV; i++) s->qvel[i] += h * qacc[i];
    for (int i = 0; i < 9; i++) s->qpos[i] += h * s->qvel[i];
    for (int i = 0; i < 3; i++) s->qpos[9 + i] += h * s->qvel[9 + i];
    /* quaternion: rotate by h*omega expressed in the body frame (right-multiply), renormalise */
    double w[3] = {s->qvel[12], s->qvel[13], s->qvel[14]}, ang = norm3(w) * h;
    if (ang > MINVAL) {
        double ax[3];
        copy3(ax, w);
        normalize3(ax);
        double sa = sin(0.5 * ang), dq[4] = {cos(0.5 * ang), ax[0] * sa, ax[1] * sa, ax[2] * sa};
        quatmul(&s->qpos[12], &s->qpos[12], dq);
    }
    quatnormalize(&s->qpos[12]);
}

void ko_step(ko_sim *s) {
    ko_forward(s);
    euler(s);
}

ko_sim *ko_sim_new(const ko_model *m, const double hand_quat[4]) {
    ko_sim *s = (ko_sim *)calloc(1, sizeof(ko_sim));
    s->m = m;
    memcpy(s->hand_quat, hand_quat, 4 * sizeof(double));
    quatnormalize(s->hand_quat);
    s->solver = 0;
    s->solver_iterations = 8;
    s->solver_tolerance = 1e-5;
    s->ncon_max = m->ngeom > 9 ? 40 : 24; /* = the product kernels' NCON_MAX (csrc/ks_model.h) */
    s->rays_enabled = 1;
    s->qpos[12] = 1.0;
    return s;
}
void ko_sim_free(ko_sim *s) { free(s); }
size_t ko_sizeof_sim(void) { return sizeof(ko_sim); }

void ko_set_state(ko_sim *s, const double *qpos, const double *qvel, const double *qacc_warmstart) {
    memcpy(s->qpos, qpos, sizeof s->qpos);
    if (qvel) memcpy(s->qvel, qvel, sizeof s->qvel); else memset(s->qvel, 0, sizeof s->qvel);
    if (qacc_warmstart) memcpy(s->qacc_warmstart, qacc_warmstart, sizeof s->qacc_warmstart);
    else memset(s->qacc_warmstart, 0, sizeof s->qacc_warmstart);
}
