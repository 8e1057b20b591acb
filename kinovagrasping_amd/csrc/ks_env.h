// ks_env.h -- per-lane env.step() / reset orchestration shared by the gfx950 kernels and the CPU
// lane check.  Mirrors KinovaGripper_Env.step (kinova_gripper_env.py:1495-1552) and
// _set_state + sim.forward() (692-703).
#pragma once
#include "ks_core.h"
#include "ks_obs.h"

namespace ks {

template <typename T> KS_HD void hand_rotation(const T* hand_quat, T* R7) {
    T q[4] = {hand_quat[0], hand_quat[1], hand_quat[2], hand_quat[3]};
    quatnormalize(q);
    quat2mat(R7, q);
}

// SnapW: callable put(k, value) writing element k of this env's snapshot
template <typename T, typename S, typename SnapW> KS_HD void write_snapshot(S scr, const T* jpos_qpos, SnapW put) {
    for (int j = 0; j < 96; j++) put(SNAP_BP + j, scr(SCR_BP + j));
    KS_UNROLL
    for (int j = 0; j < 3; j++) {
        put(SNAP_JPOS + j, jpos_qpos[j]);
        put(SNAP_JPOS + 3 + j, jpos_qpos[3 + 2 * j]);
        put(SNAP_JPOS + 6 + j, jpos_qpos[4 + 2 * j]);
    }
}

// One env.step(): action -> ctrl (constant over the frame_skip substeps), frame_skip x mj_step.
// The snapshot is what mj_forward saw at the START of the last substep: it is written as soon as that substep's kinematics
// are done (the collision / solver / Euler stages do not touch the body poses), and on_snapshot() is called right behind it -
// the stepping kernel uses that to hand the env's rangefinder rays to whoever is free a whole substep before the env is done.
template <typename T, typename S, typename SnapW, int SUBS, typename OnSnap = NoHook>
KS_HD void lane_env_step(const Model<T>& m, const Hulls<T>& hu, LaneState<T>& st, const T* hand_quat, const T* act4, S scr, Team<SUBS> team,
                         SnapW snap_put, int frame_skip, int solver_iterations, int& ncon, int& status, float* prof = nullptr,
                         T* ws = nullptr, PairWarm* warm = nullptr, OnSnap on_snapshot = OnSnap()) {
    // ws (optional, 18 reals shared by the team): where the per-step constants live; the GPU passes LDS
    T Rpalm[9], T3[9], wrist[3], R7_[9], ctrl_[NU];
    T* R7 = ws ? ws : R7_;
    T* ctrl = ws ? ws + 9 : ctrl_;
    hand_rotation(hand_quat, R7);
    mulRR(Rpalm, R7, m.geom_R[1]);
    const T zero3[3] = {0, 0, 0};
    palm_transform(Rpalm, zero3, T3, wrist);     // only the rotation feeds the controls
    action_to_ctrl(T3, act4, ctrl);
    reset_pair_words<T>(scr, team);
    for (int sub = 0; sub < frame_skip; sub++) {
        T jq[9];
        KS_UNROLL
        for (int j = 0; j < 9; j++) jq[j] = st.qpos[j];
        const bool last = sub == frame_skip - 1;
        mj_forward_step(m, hu, st.qpos, st.qvel, st.warm, ctrl, R7, scr, team, solver_iterations, true, ncon, status, prof, warm, [&]() {
            if (last) {
                if (team.sub == 0) write_snapshot<T>(scr, jq, snap_put);
                on_snapshot();
            }
        });
    }
}

// reset: state <- qpos0, zero velocities / warm start; kinematics -> snapshot
template <typename T, typename S, typename SnapW>
KS_HD void lane_reset(const Model<T>& m, LaneState<T>& st, const T* hand_quat, const T* qpos0, S scr, SnapW snap_put) {
    KS_UNROLL
    for (int i = 0; i < NQ; i++) st.qpos[i] = qpos0[i];
    KS_UNROLL
    for (int i = 0; i < NV; i++) { st.qvel[i] = 0; st.warm[i] = 0; }
    T R7[9];
    hand_rotation(hand_quat, R7);
    Kin<T> k;
    forward_kinematics(m, st.qpos, R7, k, scr);
    write_snapshot<T>(scr, st.qpos, snap_put);
}

}  // namespace ks
