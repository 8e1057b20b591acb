/* oracle/ko_env.c -- restatement of the reference env layer around mj_step.
 * TEST INFRASTRUCTURE ONLY (see ko.h).  Pinned by tests/golden/env_layer.npz, generated from the
 * reference's own Python by tools/gen_golden_env.py.
 *
 * "ENV" = /root/reference/gym-kinova-gripper/gym_kinova_gripper/envs/kinova_gripper_env.py
 * "EXPERT" = /root/reference/gym-kinova-gripper/expert_data.py
 */
#include "ko.h"
#include <math.h>
#include <string.h>

static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(double *r, const double *a, const double *b) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    r[0] = x; r[1] = y; r[2] = z;
}
static double tri_area(const double *a, const double *b) {
    double c[3];
    cross3(c, a, b);
    return sqrt(dot3(c, c)) / 2;
}
/* Tfw (row-major 4x4) applied to a point */
static void xform(const double *T, const double *p, double *out) {
    for (int i = 0; i < 3; i++) out[i] = T[4 * i] * p[0] + T[4 * i + 1] * p[1] + T[4 * i + 2] * p[2] + T[4 * i + 3];
}

/* ENV:274-288 _get_trans_mat_wrist_pose: T = (R_palm C)^T, wrist = p_palm + T^T [-0.009,0.048,0],
 * Tfw = [T, -T wrist] */
void ko_env_palm_transform(const double palm_xpos[3], const double palm_xmat[9], double Tfw[16], double wrist[3]) {
    static const double C[9] = {0, 0, 1, -1, 0, 0, 0, -1, 0};
    double RC[9], T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            RC[3 * i + j] = palm_xmat[3 * i] * C[j] + palm_xmat[3 * i + 1] * C[3 + j] + palm_xmat[3 * i + 2] * C[6 + j];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[3 * i + j] = RC[3 * j + i];
    const double off[3] = {-0.009, 0.048, 0.0};
    for (int i = 0; i < 3; i++) wrist[i] = palm_xpos[i] + T[i] * off[0] + T[3 + i] * off[1] + T[6 + i] * off[2];
    memset(Tfw, 0, 16 * sizeof(double));
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Tfw[4 * i + j] = T[3 * i + j];
        Tfw[4 * i + 3] = -(T[3 * i] * wrist[0] + T[3 * i + 1] * wrist[1] + T[3 * i + 2] * wrist[2]);
    }
    Tfw[15] = 1;
}

/* ENV:1495-1535: action -> 9 actuator controls (constant over the 15 substeps) */
void ko_env_ctrl(const double Tfw[16], const double *action, int naction, double ctrl[9]) {
    double a[6];
    if (naction == 4) { a[0] = 0; a[1] = 0; for (int i = 0; i < 4; i++) a[2 + i] = action[i]; }
    else for (int i = 0; i < 6; i++) a[i] = action[i];
    const double ff = 0.733 * 10 / 25; /* ENV:1509-1511 mass*10/gear */
    double stuff[3], slide[3];
    for (int i = 0; i < 3; i++) {
        stuff[i] = Tfw[4 * i + 2] * ff;
        slide[i] = Tfw[4 * i] * a[0] + Tfw[4 * i + 1] * a[1] + Tfw[4 * i + 2] * a[2];
    }
    stuff[0] = -stuff[0]; stuff[1] = -stuff[1];
    slide[0] = -slide[0]; slide[1] = -slide[1]; /* ENV:1519-1522: both branches identical */
    for (int i = 0; i < 3; i++) {
        ctrl[2 * i] = slide[i];
        ctrl[6 + i] = a[3 + i];
        ctrl[2 * i + 1] = stuff[i];
    }
}

/* ENV:591-608 _get_dot_product */
static double dot_product20(const double *obj, const double *hand) {
    double ox = fabs(obj[0] - hand[0]), oy = fabs(obj[1] - hand[1]);
    double on = sqrt(ox * ox + oy * oy);
    double cx = fabs(0.0 - hand[0]), cy = fabs(0.0 - hand[1]);
    double cn = sqrt(cx * cx + cy * cy);
    double d = (ox / on) * (cx / cn) + (oy / on) * (cy / cn);
    return pow(d, 20);
}

typedef struct {
    double Tfw[16], wrist[3], x_angle, z_angle, joint_states[9], fo_dist[12], range[17], dot_obj, finger_dot[6];
} common_t;

static void obs_common(const ko_env_inputs *in, common_t *c) {
    ko_env_palm_transform(in->palm_xpos, in->palm_xmat, c->Tfw, c->wrist);
    /* ENV:563-582 _get_angles */
    double lo[3];
    xform(c->Tfw, in->obj_xpos, lo);
    double n = sqrt(dot3(lo, lo)), ow[3] = {lo[0] / n, lo[1] / n, lo[2] / n};
    c->z_angle = acos(ow[1] / sqrt(ow[0] * ow[0] + ow[1] * ow[1]));
    c->x_angle = acos(ow[1] / sqrt(ow[1] * ow[1] + ow[2] * ow[2]));
    /* ENV:356-362 */
    for (int i = 0; i < 9; i++) c->joint_states[i] = in->sensordata[i];
    c->joint_states[0] = -c->joint_states[0];
    c->joint_states[1] = -c->joint_states[1];
    /* ENV:538-548: site order f1_prox f1_prox_1 f2_prox f2_prox_1 f3_prox f3_prox_1 f1_dist f1_dist_1 ...
     * model site order: palm x5, then per finger prox, prox_1, dist, dist_1 */
    static const int order[12] = {5, 6, 9, 10, 13, 14, 7, 8, 11, 12, 15, 16};
    for (int k = 0; k < 12; k++) {
        const double *p = in->site_xpos[order[k]];
        double d[3] = {fabs(p[0] - in->obj_xpos[0]), fabs(p[1] - in->obj_xpos[1]), fabs(p[2] - in->obj_xpos[2])};
        c->fo_dist[k] = sqrt(dot3(d, d));
    }
    /* ENV:552-561 */
    for (int i = 0; i < 17; i++) c->range[i] = in->sensordata[9 + i] == -1 ? 6 : in->sensordata[9 + i];
    c->dot_obj = dot_product20(in->obj_xpos, in->link7_xpos);
    for (int k = 0; k < 6; k++) c->finger_dot[k] = dot_product20(in->finger_xpos[k], in->link7_xpos);
}

/* ENV:438-534, state_rep == "local" */
void ko_env_obs_local(const ko_env_inputs *in, double obs[KO_NOBS]) {
    common_t c;
    obs_common(in, &c);
    double fp[18];
    for (int k = 0; k < 6; k++) xform(c.Tfw, in->finger_xpos[k], &fp[3 * k]);
    memcpy(obs, fp, sizeof fp);
    xform(c.Tfw, c.wrist, &obs[18]);
    xform(c.Tfw, in->obj_xpos, &obs[21]);
    memcpy(&obs[24], c.joint_states, sizeof c.joint_states);
    obs[33] = in->obj_size[0]; obs[34] = in->obj_size[1]; obs[35] = in->obj_size[2] * 2;
    memcpy(&obs[36], c.fo_dist, sizeof c.fo_dist);
    obs[48] = c.x_angle; obs[49] = c.z_angle;
    memcpy(&obs[50], c.range, sizeof c.range);
    double g[3] = {-c.Tfw[2], -c.Tfw[6], -c.Tfw[10]}; /* Tfw[:3,:3] @ [0,0,-1] */
    obs[67] = g[0]; obs[68] = g[1]; obs[69] = g[2];
    /* ENV:290-343 experimental_sensor */
    double s1[3], s2[3];
    for (int i = 0; i < 3; i++) { s1[i] = fp[i] - fp[6 + i]; s2[i] = fp[i] - fp[3 + i]; }
    double front_area = tri_area(s1, s2);
    double top1 = tri_area(&fp[0], &fp[9]), top2 = tri_area(&fp[9], &fp[12]), top3 = tri_area(&fp[3], &fp[12]);
    double top4 = tri_area(&fp[6], &fp[15]), top5 = tri_area(&fp[9], &fp[15]);
    double total1 = top1 + top2 + top3, total2 = top1 + top4 + top5, top_area = total1 > total2 ? total1 : total2;
    double sx = 0, sy = 0, sz = 0;
    int nh = 0;
    for (int i = 0; i < 5; i++) {
        if (c.range[i] < 0.06) {
            double t[3];
            xform(c.Tfw, in->site_xpos[i], t);
            t[1] += c.range[i];
            sx += t[0]; sy += t[1]; sz += t[2];
            nh++;
        }
    }
    if (nh == 0) { obs[70] = obs[71] = obs[72] = 0.2; }
    else { obs[70] = sx / nh; obs[71] = sy / nh; obs[72] = sz / nh; }
    const double *sz3 = in->obj_size;
    int am = 0;
    if (fabs(g[1]) > fabs(g[am])) am = 1;
    if (fabs(g[2]) > fabs(g[am])) am = 2;
    double front_part, top_part;
    if (am == 2) { front_part = fabs(sz3[0] * sz3[2]) / front_area; top_part = fabs(sz3[0] * sz3[1]) / top_area; }
    else if (am == 1) { front_part = fabs(sz3[0] * sz3[2]) / front_area; top_part = fabs(sz3[1] * sz3[2]) / top_area; }
    else { front_part = fabs(sz3[0] * sz3[1]) / front_area; top_part = fabs(sz3[0] * sz3[2]) / top_area; }
    obs[73] = front_part; obs[74] = top_part;
    memcpy(&obs[75], c.finger_dot, sizeof c.finger_dot);
    obs[81] = c.dot_obj;
}

/* ENV:486-494, state_rep == "global" (74 values; only obs[23] feeds the reward) */
void ko_env_obs_global(const ko_env_inputs *in, double obs[KO_NOBS_GLOBAL]) {
    common_t c;
    obs_common(in, &c);
    for (int k = 0; k < 6; k++) memcpy(&obs[3 * k], in->finger_xpos[k], 3 * sizeof(double));
    memcpy(&obs[18], c.wrist, 3 * sizeof(double));
    memcpy(&obs[21], in->obj_xpos, 3 * sizeof(double));
    memcpy(&obs[24], c.joint_states, sizeof c.joint_states);
    obs[33] = in->obj_size[0]; obs[34] = in->obj_size[1]; obs[35] = in->obj_size[2] * 2;
    memcpy(&obs[36], c.fo_dist, sizeof c.fo_dist);
    obs[48] = c.x_angle; obs[49] = c.z_angle;
    memcpy(&obs[50], c.range, sizeof c.range);
    memcpy(&obs[67], c.finger_dot, sizeof c.finger_dot);
    obs[73] = c.dot_obj;
}

/* ENV:631-687 _get_reward with with_grasp_reward=False */
void ko_env_reward(double obj_z_world, double *reward, int *done, double info[3]) {
    const double target = 0.2;
    double lift = 0;
    *done = 0;
    if (fabs(obj_z_world - target) < 0.005 || obj_z_world >= target) { lift = 50.0; *done = 1; }
    info[0] = 0.0; info[1] = 0.0; info[2] = lift; /* finger, grasp, lift */
    *reward = 0.2 * 0.0 + lift + 0.0;
}

/* EXPERT:559-593; arguments are obs[9:17] of the previous and current observation */
int ko_check_grasp(const double *f_dist_old, const double *f_dist_new) {
    const double sampling_time = 15;
    double total = fabs(f_dist_old[0] - f_dist_new[0]) / sampling_time + fabs(f_dist_old[3] - f_dist_new[3]) / sampling_time +
                   fabs(f_dist_old[6] - f_dist_new[6]) / sampling_time;
    return total < 0.0002;
}

void ko_env_inputs_from_sim(const ko_sim *s, ko_env_inputs *in) {
    static const int fg[6] = {2, 4, 6, 3, 5, 7}; /* f1_prox f2_prox f3_prox f1_dist f2_dist f3_dist */
    memcpy(in->palm_xpos, s->geom_xpos[1], sizeof in->palm_xpos);
    memcpy(in->palm_xmat, s->geom_xmat[1], sizeof in->palm_xmat);
    for (int k = 0; k < 6; k++) memcpy(in->finger_xpos[k], s->geom_xpos[fg[k]], 3 * sizeof(double));
    memcpy(in->obj_xpos, s->geom_xpos[8], sizeof in->obj_xpos);
    memcpy(in->link7_xpos, s->xpos[2], sizeof in->link7_xpos);
    memcpy(in->site_xpos, s->site_xpos, sizeof in->site_xpos);
    memcpy(in->sensordata, s->sensordata, sizeof in->sensordata);
    /* obj_size_obs = [s0, s1, 2*s2]; the env's _get_obj_size returns the value before the x2 */
    in->obj_size[0] = s->m->obj_size_obs[0];
    in->obj_size[1] = s->m->obj_size_obs[1];
    in->obj_size[2] = s->m->obj_size_obs[2] / 2;
}

/* ENV:1495-1552.  Note the reference never calls sim.forward() after the 15th sim.step(), so
 * every quantity the observation reads (geom/site poses, sensordata) is the one mj_forward
 * computed at the START of the last substep. */
void ko_env_step(ko_sim *s, const double *action, int naction, int frame_skip, double obs[KO_NOBS],
                 double *reward, int *done, double info[3]) {
    double Tfw[16], wrist[3];
    ko_env_palm_transform(s->geom_xpos[1], s->geom_xmat[1], Tfw, wrist);
    ko_env_ctrl(Tfw, action, naction, s->ctrl);
    /* MuJoCo evaluates the 17 rangefinders every substep; only the last substep's values are read, so the
     * oracle skips the rest (same result, the cpu_baseline is correspondingly generous to the CPU) */
    int keep = s->rays_enabled;
    for (int k = 0; k < frame_skip; k++) { s->rays_enabled = keep && (k == frame_skip - 1); ko_step(s); }
    s->rays_enabled = keep;
    ko_env_inputs in;
    ko_env_inputs_from_sim(s, &in);
    ko_env_obs_local(&in, obs);
    ko_env_reward(in.obj_xpos[2], reward, done, info);
}

/* ENV:692-703 _set_state (+ fresh MjSim: zero velocities) followed by _get_obs (ENV:1381) */
void ko_env_reset(ko_sim *s, const double qpos0[KO_NQ], double obs[KO_NOBS]) {
    ko_set_state(s, qpos0, NULL, NULL);
    memset(s->ctrl, 0, sizeof s->ctrl);
    ko_forward(s);
    if (obs) {
        ko_env_inputs in;
        ko_env_inputs_from_sim(s, &in);
        ko_env_obs_local(&in, obs);
    }
}
