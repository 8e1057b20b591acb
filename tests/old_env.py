"""TEST INFRASTRUCTURE: the reference's OLD env (gym_kinova_gripper/envs/kinova_gripper_env_s.py on
`kinova_description/j2s7s300_end_effector_v1_mbox (copy).xml`) emulated on the compiled `mbox` model, so that the one contact
trajectory of real MuJoCo 1.50 the reference tree holds (Old Code/Pose_file_2.csv, fixture tests/golden/mujoco_recorded.npz:
pose_file_2) can be replayed by the oracle and by the HIP kernels.

What differs between the old XML and today's `..._mbox.xml` (diffed line by line): ONE slide joint `j2s7s300_joint_7` (axis 1 0 0
= today's slide_z, range 0 - 0.2) instead of three, four velocity actuators (wrist kv 150 with ctrlrange +-0.2, three fingers
kv 2.5 with ctrlrange +-0.8) instead of nine, no gravity feed-forward motors.  Everything else - bodies, meshes, box, contact
pairs, tendons, defaults - is identical.  Emulation on the compiled blob: the x / y slides get an armature of 1e10 kg (they do
not move and drop out of every inverse weight, exactly as if the joints did not exist), `dof / body / tendon_invweight0` are
recomputed for that model (they enter the constraint regularisation R), the slide servo's ctrlrange becomes 0.2 and slide_z's
range 0 - 0.2; the motors stay un-commanded.  The old env's step (kinova_gripper_env_s.py:681-696): frame_skip 4,
ctrl[0] = action[0] / 0.8 * 0.2 (0 if negative), ctrl[1:4] = action[1:4] (0 if negative).

The recording holds no actions (a policy network produced them, Old Code/main_DDPGfD_OG.py:36-56).  They are recovered row by
row from the four ACTUATED joint angles the recording does hold: the command of row r is constant over its 4 substeps and
row r's sensors are evaluated in the forward pass of the 4th, i.e. after 3 integrations under it - a square, monotone, almost
diagonal 4 x 4 problem per row (velocity servos) - until the box is grasped, and not one-to-one where a finger rubs on the box while
the wrist servo lifts the hand (`replay_recording`) - solved by a bounded Newton iteration on the oracle.  The other 44 columns of a row
are then predictions: the three distal joints (soft tendons under contact load), the object's pushed path, six finger-link
centres, the palm, 13 site-object distances, the dot product."""
from __future__ import annotations

import numpy as np

from kinovagrasping_amd import model_compiler as mc, scenarios

ROW_SITES = ["palm_1", "f1_prox", "f1_prox_1", "f2_prox", "f2_prox_1", "f3_prox", "f3_prox_1", "f1_dist", "f1_dist_1", "f2_dist", "f2_dist_1",
             "f3_dist", "f3_dist_1"]                                       # kinova_gripper_env_s.py:212
_SITE_IDX = [mc.SITE_NAMES.index(n) for n in ROW_SITES]
U_LO, U_HI = np.zeros(4), np.array([0.2, 0.8, 0.8, 0.8])                   # wrist servo ctrlrange 0.2, finger servos 0.8; negatives are zeroed
FRAME_SKIP = 4


def old_env_blob(edit=None) -> bytes:
    """`mbox` compiled model edited into `..._mbox (copy).xml` (see the module docstring)"""
    M = mc.read_blob(scenarios.model_blob("mbox"))
    M["dof_armature"] = M["dof_armature"].copy()
    M["dof_armature"][0:2] = 1e10
    M.update(mc._invweights(M))
    M["actuator"] = M["actuator"].copy()
    M["actuator"][2] = 0.2
    M["slide_range"] = M["slide_range"].copy()
    M["slide_range"][2] = [0.0, 0.2]
    if edit is not None:
        edit(M)
    return mc.blob_bytes(M)


def ctrl_of(u):
    """the old env's four controls [wrist, f1, f2, f3] in today's nine-actuator layout"""
    c = np.zeros(9)
    c[4] = u[0]
    c[6:9] = u[1:4]
    return c


def start_qpos(row0):
    q = np.zeros(16)
    q[9:12] = row0[21:24]           # the recorded release point (0.055, 0, 0.05000025): 5 mm inside the floor
    q[12] = 1.0
    return q


def row_from_kinematics(geom_xpos, site_xpos, link7_xpos, jointpos):
    """the 48 recorded columns (kinova_gripper_env_s.py:181-209, state_rep "global") from one forward pass:
    geom_xpos [9, 3], site_xpos [17, 3], link7_xpos [3], jointpos sensors [9] (slide x y z, three proximal, three distal)"""
    gx, sx, sd = np.asarray(geom_xpos).reshape(-1, 3), np.asarray(site_xpos).reshape(17, 3), np.asarray(jointpos)
    obj = gx[8]
    dists = []
    for i in _SITE_IDX:                                                      # _get_finger_obj_dist, :210-224
        d = np.abs(sx[i][:2] - obj[:2])
        d[0] -= 0.0175
        dists.append(np.linalg.norm(d))
    ov, cv = np.abs(obj[:2] - link7_xpos[:2]), np.abs(-np.asarray(link7_xpos)[:2])   # _get_dot_product, :266-281
    dot = float((ov / np.linalg.norm(ov)) @ (cv / np.linalg.norm(cv))) ** 20
    return np.concatenate([gx[[2, 4, 6, 3, 5, 7]].ravel(), gx[1], obj, sd[2:9], [0.02125, 0.02125, 0.11], dists, [dot]])


def oracle_row(s):
    """row of the forward pass the oracle sim `s` ran last (ko_step = forward + integrate: its views hold the pre-integration
    kinematics and sensors - what MuJoCo's sensordata / geom_xpos hold after mj_step)"""
    return row_from_kinematics(s.view("geom_xpos"), s.view("site_xpos"), s.view("xpos").reshape(10, 3)[2], s.view("sensordata")[:9])


def oracle_state(s):
    return s.view("qpos").copy(), s.view("qvel").copy(), s.view("qacc_warmstart").copy()


def new_oracle_sim(edit=None, solver_iterations=100):
    from oracle import ko_py as ko
    m = ko.OracleModel(old_env_blob(edit))
    s = ko.OracleSim(m, scenarios.hand_quat_for("normal"), solver_iterations=solver_iterations)
    s.s.rays_enabled = 0
    s._keep_model = m
    return s


PREDICTED_COLS = [c for c in range(47) if c not in (24, 25, 26, 27, 31, 32, 33)]   # everything but the 4 actuated joints and the constant box size


def _run_row(s, st, u):
    s.set_state(*st)
    for _ in range(FRAME_SKIP):
        s.step(ctrl_of(u))
    return oracle_row(s)


def recover_commands(s, st, tgt, u, iters=14):
    """Newton iteration on the row's four commands for the four actuated joint angles `tgt` (recording columns 24-27), from the
    start `u`: FULL 4 x 4 finite-difference Jacobian (once the box is grasped the fingers couple through it: a diagonal
    iteration stalls at 1e-6), commands that want to leave their range are held at the bound and the others re-solved."""
    u = np.array(u, dtype=np.float64)
    for _ in range(iters):
        row = _run_row(s, st, u)
        res = row[24:28] - tgt
        if np.abs(res).max() < 1e-12:
            break
        J = np.zeros((4, 4))
        for k in range(4):
            h = 1e-6 if u[k] < U_HI[k] - 1e-6 else -1e-6
            u2 = u.copy()
            u2[k] += h
            J[:, k] = (_run_row(s, st, u2)[24:28] - row[24:28]) / h
        free, du = np.ones(4, bool), np.zeros(4)
        for _ in range(4):
            du = np.zeros(4)
            du[free] = np.linalg.lstsq(J[:, free], -res, rcond=None)[0]
            out = free & ((u + du < U_LO - 1e-15) | (u + du > U_HI + 1e-15))
            if not out.any():
                break
            free &= ~out
        un = np.clip(u + du, U_LO, U_HI)
        if np.abs(un - u).max() < 1e-14:
            break
        u = un
    return u, _run_row(s, st, u)


def replay_recording(rec, s=None, n_rows=None, branch_tol=1e-8, grid=17):
    """Replays pose_file_2 on the oracle: returns (rows [R, 48], controls [R, 4], states) where states[r] = (qpos, qvel, warm)
    after the 4 r integrations of rows 1..r; rows[0] is the reset row.  controls[r] reproduce the four actuated joint angles of
    row r (columns 24-27) wherever a command inside the servo ranges can.

    The map command -> actuated joint angle is NOT one-to-one (round 5; rounds 3-4 assumed it was and left the recording at row
    22): the wrist servo carries the hand UP while finger 1 rubs on the box, the friction of that contact (up to 1 N on a 0.8 kg
    hand) changes direction with the sign of the relative vertical velocity, and two wrist commands - 0.0382 and 0.0512 in row
    22 - put the wrist slide at the SAME recorded position after the row's three integrations, with the friction pointing up in
    one and down in the other.  The other 44 columns tell them apart (the box: 1.8e-4 against 1e-10).  So: Newton from the
    previous row's commands; if the predicted columns then miss the recording by more than `branch_tol`, restart it from a grid
    over each command's range and keep the solution whose actuated AND predicted columns are closest to the recording.  The
    selection uses the recording only to choose between solutions of the 4 x 4 problem; nothing but the 4 commands is fitted."""
    if s is None:
        s = new_oracle_sim()
    n_rows = len(rec) if n_rows is None else n_rows
    s.set_state(start_qpos(rec[0]))
    s.forward()
    rows, us, states = [oracle_row(s)], [np.zeros(4)], [oracle_state(s)]
    u = np.array([0.0, 0.8, 0.0, 0.8])

    def miss(row, r):
        return np.abs(row[24:28] - rec[r, 24:28]).max() + np.abs(row[PREDICTED_COLS] - rec[r, PREDICTED_COLS]).max()

    for r in range(1, n_rows):
        st, tgt = states[-1], rec[r, 24:28]
        best_u, best_row = recover_commands(s, st, tgt, u)
        best = miss(best_row, r)
        if best > branch_tol:
            u1 = best_u.copy()
            for k in range(4):
                for v in np.linspace(U_LO[k], U_HI[k], grid):
                    u0 = u1.copy()
                    u0[k] = v
                    u2, row2 = recover_commands(s, st, tgt, u0, iters=8)
                    e = miss(row2, r)
                    if e < best:
                        best, best_u, best_row = e, u2, row2
                    if best <= branch_tol:
                        break
                if best <= branch_tol:
                    break
        u = best_u
        rows.append(_run_row(s, st, u))
        us.append(u.copy())
        states.append(oracle_state(s))
    return np.array(rows), np.array(us), states
