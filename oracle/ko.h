/* oracle/ko.h -- CPU oracle for the Kinova gripper hot path (TEST INFRASTRUCTURE ONLY).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (kinovagrasping_amd/csrc) never includes, links or calls anything here.
 *
 * What it restates (reference = /root/reference/gym-kinova-gripper/gym_kinova_gripper/envs/
 * kinova_gripper_env.py, "ENV"):
 *   - env layer, PINNED by tests/golden/env_layer.npz generated from the reference's own
 *     Python (tools/gen_golden_env.py): action->ctrl mapping ENV:1495-1535, palm transform
 *     ENV:274-288, 82-d / 74-d observation ENV:438-534 (+ helpers ENV:290-343, 356-362,
 *     538-608), reward/termination ENV:631-687.
 *   - physics (mj_step, called at ENV:1535): lives in MuJoCo 1.50 / mujoco-py 1.50.1.0, a
 *     third-party binary that is NOT under /root/reference and not installable here.  The
 *     restatement follows MuJoCo's published algorithm (Computation chapter) on the compiled
 *     model data.  The reference's TESTS hold no golden vectors at the mj_step boundary (SURVEY.md
 *     sec. 4, 8c); its tree does hold four sets of recorded MuJoCo 1.50 output (a 63-row x 48-column
 *     contact trajectory - box pushed, grasped and lifted -, finger joint traces, ten demonstrations,
 *     success / failure maps): the physics is PINNED to those (tests/test_mujoco_recorded.py,
 *     the contact trajectory, replayed free-running, to 1.05e-10 in every column
 *     through row 40 - push, grasp, first centimetres of the lift -, 8e-8 through row 45, 5.5e-4 for the
 *     rest of the lift; ko_physics.c's header has the list).  Contact forces and velocities of real
 *     MuJoCo were never recorded.
 *   - multi-geom objects (Bottle / TBottle / Bowl / RBowl: `object` + jointless child bodies, ..._sbottle.xml:158-186): PARITY
 *     UNPINNED for what is specific to them - the reference tree holds no MuJoCo output of these models.  The restatement follows
 *     MuJoCo's documented semantics (welded children = one rigid body with the composite inertial; their geoms collide dynamically
 *     at the geom margin / default friction, never with each other; a contact's regularisation uses the inverse weight of the body
 *     that owns the geom) on top of the single-geom pipeline that IS pinned.
 */
#ifndef KO_H
#define KO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KO_NQ 16
#define KO_NV 15
#define KO_NU 9
#define KO_NBODY 10
#define KO_NGEOM 17   /* capacity: ground, 7 hand geoms, `object` + up to 8 welded pieces (multi-geom objects); ko_model.ngeom is the count */
#define KO_OBJ_GEOM 8 /* the geom named `object` */
#define KO_NMESH 12   /* capacity: palm, proximal, distal + one hull per object geom; ko_model.nmesh */
#define KO_NSITE 17
#define KO_NSENSOR 26
#define KO_NPAIR_MAX 96
#define KO_NCON_MAX 40 /* capacity; ko_sim.ncon_max = contacts kept per substep: 24 (nine-geom models), 40 (multi-geom objects) */
#define KO_NEFC_MAX (3 + 9 + 4 * KO_NCON_MAX)
#define KO_NOBS 82
#define KO_NOBS_GLOBAL 74

typedef struct {
    double dt, impratio, gravity_z, margin, solref[2], solimp[3], mpr_tolerance;
    int mpr_iterations;
    double body_pos[KO_NBODY][3], body_quat[KO_NBODY][4], body_mass[KO_NBODY];
    double body_ipos[KO_NBODY][3], body_iquat[KO_NBODY][4], body_inertia[KO_NBODY][3];
    double slide_axis[3][3], slide_range[3][2], hinge_range[6][2];
    int hinge_limited[6];
    double dof_damping[KO_NV], dof_armature[KO_NV];
    double geom_pos[KO_NGEOM][3], geom_quat[KO_NGEOM][4], geom_size[KO_NGEOM][3], geom_rbound[KO_NGEOM];
    int geom_body[KO_NGEOM], geom_mesh[KO_NGEOM];
    int ngeom, nmesh;
    int mesh_nvert[KO_NMESH], mesh_ntri[KO_NMESH];
    double *mesh_vert[KO_NMESH], *mesh_tri[KO_NMESH]; /* hull vertices [nvert][3]; original triangles [ntri][9], geom frame */
    double site_pos[KO_NSITE][3], site_quat[KO_NSITE][4];
    int site_body[KO_NSITE];
    int npair;
    double pairs[KO_NPAIR_MAX][5]; /* g1, g2, mu1, mu2, margin */
    double tendon_coef[3][2];
    double actuator[5]; /* kv_slide, gear_motor, ctrlrange_slide, kv_finger, ctrlrange_finger */
    double dof_invweight0[KO_NV], body_invweight0[KO_NBODY][2], tendon_invweight0[3];
    /* translational inverse weight of the MuJoCo body that owns geom g: body_invweight0[geom_body[g]][0], except for the pieces of a
     * multi-geom object - MuJoCo keeps each welded piece as a body of its own with its own value (blob record geom_invweight0) */
    double geom_invweight0[KO_NGEOM];
    double obj_size_obs[3];
} ko_model;

typedef struct {
    double dist, pos[3], frame[9], mu[2], margin;
    int geom1, geom2;
} ko_contact;

typedef struct {
    const ko_model *m;
    double hand_quat[4];
    int solver;            /* 0 = Newton (default, MuJoCo's default), 1 = PGS (comparison) */
    int solver_iterations; /* fixed iteration count: Newton steps or PGS sweeps */
    int ncon_max;
    /* state */
    double qpos[KO_NQ], qvel[KO_NV], qacc_warmstart[KO_NV], ctrl[KO_NU];
    /* position-dependent */
    double xpos[KO_NBODY][3], xmat[KO_NBODY][9], xipos[KO_NBODY][3], ximat[KO_NBODY][9];
    double geom_xpos[KO_NGEOM][3], geom_xmat[KO_NGEOM][9];
    double site_xpos[KO_NSITE][3], site_xmat[KO_NSITE][9];
    double M[KO_NV][KO_NV], L[KO_NV][KO_NV];
    int ncon, ncon_dropped;
    ko_contact contact[KO_NCON_MAX];
    int nefc;
    int efc_type[KO_NEFC_MAX]; /* 0 equality, 1 limit, 2 contact */
    double efc_J[KO_NEFC_MAX][KO_NV], efc_pos[KO_NEFC_MAX], efc_margin[KO_NEFC_MAX];
    double efc_R[KO_NEFC_MAX], efc_aref[KO_NEFC_MAX], efc_b[KO_NEFC_MAX], efc_force[KO_NEFC_MAX];
    double sensordata[KO_NSENSOR];
    /* velocity / force */
    double qfrc_bias[KO_NV], qfrc_passive[KO_NV], qfrc_actuator[KO_NV], qfrc_constraint[KO_NV];
    double qacc_smooth[KO_NV], qacc[KO_NV];
    /* env layer bookkeeping */
    int mpr_calls, mpr_support_calls;
    double newton_last_grad;
    int newton_iters_used;
    int rays_enabled; /* 0: skip the 17 rangefinders in ko_forward (only the last substep's values are ever read) */
    /* per-env domain randomisation (BASELINE config 5; an extension - the reference fixes mass 0.1 and mu 1):
     * obj_mass > 0 replaces the object's mass (inertia scales with it, its body_invweight0 by
     * (m0 + armature) / (m + armature)); obj_mu > 0 replaces the friction of the object-hand pairs. */
    double obj_mass, obj_mu;
    /* Newton stop rule: the iteration ends when the last step is below solver_tolerance * (1 + |qacc|_inf) (1e-5 by default, the
     * product kernels' rule) or at solver_iterations.  newton_converged = 1 when the rule fired (0: the cap ended the loop),
     * newton_last_step = that last relative step - what the solver-cap study (tests/studies/solver_cap.py) reads. */
    double solver_tolerance, newton_last_step;
    int newton_converged;
    int narrow_phase; /* 0 (default, = the product kernels): GJK closest features in the margin zone, MPR on overlap; 1 (study only):
                       * MPR on margin-inflated hulls for both, MuJoCo 1.50's scheme */
} ko_sim;

/* ---- model ---- */
ko_model *ko_model_load(const void *blob, size_t nbytes);
void ko_model_free(ko_model *m);

/* ---- physics (S0-S8) ---- */
ko_sim *ko_sim_new(const ko_model *m, const double hand_quat[4]);
void ko_sim_free(ko_sim *s);
void ko_set_state(ko_sim *s, const double *qpos, const double *qvel, const double *qacc_warmstart);
void ko_forward(ko_sim *s); /* mj_forward */
void ko_step(ko_sim *s);    /* mj_step = forward + Euler */
void ko_kinematics(ko_sim *s);

/* ---- env layer (E1, O1-O3, L5) ---- */
typedef struct { /* what the reference reads from mujoco-py (fake-sim inputs of the golden vectors) */
    double palm_xpos[3], palm_xmat[9];
    double finger_xpos[6][3]; /* f1_prox f2_prox f3_prox f1_dist f2_dist f3_dist */
    double obj_xpos[3];
    double link7_xpos[3];
    double site_xpos[KO_NSITE][3]; /* model site order */
    double sensordata[KO_NSENSOR];
    double obj_size[3]; /* _get_obj_size(): [s0, s1, s2] before the x2 */
} ko_env_inputs;

void ko_env_palm_transform(const double palm_xpos[3], const double palm_xmat[9], double Tfw[16], double wrist[3]);
void ko_env_ctrl(const double Tfw[16], const double *action, int naction, double ctrl[9]);
void ko_env_obs_local(const ko_env_inputs *in, double obs[KO_NOBS]);
void ko_env_obs_global(const ko_env_inputs *in, double obs[KO_NOBS_GLOBAL]);
void ko_env_reward(double obj_z_world, double *reward, int *done, double info[3]);
int ko_check_grasp(const double *f_dist_old, const double *f_dist_new);

void ko_env_inputs_from_sim(const ko_sim *s, ko_env_inputs *in);
/* full env.step(): 15 x mj_step then obs/reward exactly as ENV:1495-1552 */
void ko_env_step(ko_sim *s, const double *action, int naction, int frame_skip, double obs[KO_NOBS],
                 double *reward, int *done, double info[3]);
void ko_env_reset(ko_sim *s, const double qpos0[KO_NQ], double obs[KO_NOBS]);

size_t ko_sizeof_sim(void);

#ifdef __cplusplus
}
#endif
#endif
