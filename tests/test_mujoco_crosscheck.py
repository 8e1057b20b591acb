"""Cross-check against REAL MuJoCo, for machines that have the third-party `mujoco` Python package (this image and the
GPU boxes do not: the whole module is skipped there - SURVEY hard part 1c).

Nothing of the reference is read: the model is rebuilt as a self-contained MJCF from the compiled blob
(kinovagrasping_amd/mjcf_export.py).  What is compared, strongest first:
  * the joint-space inertia matrix at qpos0 (mj_fullM) against the blob's M0 / the oracle's M: 1e-9 relative;
  * contact-free motion (object in free flight, hand hovering on its servos): 1e-9 per step - pins armature, implicit
    damping, actuator models, gravity compensation;
  * BASELINE config 1 (30 env-steps of PCG64(0) actions on a CubeS): the first substep whose relative qpos error
    exceeds 1e-4 and the phase it is in are REPORTED; asserted only for the first 20 substeps (resting contact on the
    plane), because a modern MuJoCo is not MuJoCo 1.50 (native convex collision, multi-contact defaults).
The CPU half (always run) checks that the exported MJCF is well formed and complete."""
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from kinovagrasping_amd import mjcf_export, scenarios
from kinovagrasping_amd.model_compiler import read_blob


def test_exported_mjcf_is_complete():
    blob = scenarios.model_blob("CubeS")
    root = ET.fromstring(mjcf_export.to_mjcf(blob, scenarios.hand_quat_for("normal")))
    M = read_blob(blob)
    assert len(root.findall(".//worldbody//body")) == 9 and len(root.findall(".//worldbody//geom")) == 9
    assert len(root.findall(".//site")) == 17 and len(root.findall(".//contact/pair")) == len(M["pairs"]) == 30
    assert len(root.findall(".//worldbody//joint")) == 10 and len(root.findall(".//actuator/*")) == 9
    assert len(root.findall(".//sensor/jointpos")) == 9 and len(root.findall(".//sensor/rangefinder")) == 17
    assert root.find("compiler").get("autolimits") == "false"            # tendon `range` must stay inactive as in 1.50
    masses = [float(b.find("inertial").get("mass")) for b in root.findall(".//worldbody//body") if b.find("inertial") is not None]
    assert abs(sum(masses) - (0.727 + 6 * 0.01 + 0.1)) < 1e-12
    nv = [len(m.get("vertex").split()) // 3 for m in root.findall(".//asset/mesh")]
    assert nv == [len(M[f"mesh{k}_vert"]) for k in range(4)]


def _mj():
    return pytest.importorskip("mujoco", reason="third-party mujoco package not installed (expected on this image)")


def _load(shape="CubeS", orientation="normal"):
    mujoco = _mj()
    xml = mjcf_export.to_mjcf(scenarios.model_blob(shape), scenarios.hand_quat_for(orientation))
    m = mujoco.MjModel.from_xml_string(xml)
    return mujoco, m, mujoco.MjData(m)


def test_mass_matrix_matches_real_mujoco():
    from oracle import ko_py as ko
    mujoco, m, d = _load()
    assert (m.nq, m.nv, m.nu, m.nsensordata) == (16, 15, 9, 26)
    mujoco.mj_forward(m, d)
    Mfull = np.zeros((m.nv, m.nv))
    mujoco.mj_fullM(m, Mfull, d.qM)
    M0 = read_blob(scenarios.model_blob("CubeS"))["M0"]
    np.testing.assert_allclose(Mfull, M0, rtol=1e-9, atol=1e-12)
    o = ko.OracleSim(ko.OracleModel(scenarios.model_blob("CubeS")), scenarios.hand_quat_for("normal"))
    o.set_state(d.qpos.copy()); o.forward()
    np.testing.assert_allclose(o.view("M").reshape(15, 15), Mfull, rtol=1e-9, atol=1e-12)


def test_contact_free_motion_matches_real_mujoco():
    from oracle import ko_py as ko
    mujoco, m, d = _load()
    o = ko.OracleSim(ko.OracleModel(scenarios.model_blob("CubeS")), scenarios.hand_quat_for("normal"), solver_iterations=50)
    q0 = np.zeros(16); q0[2] = 0.3; q0[9:12] = [0.0, -0.3, 3.0]; q0[12] = 1          # hand lifted off the floor, object in flight
    ctrl = np.zeros(9); ctrl[5] = 0.2932; ctrl[6:9] = [0.4, -0.2, 0.3]; ctrl[0] = 0.05
    d.qpos[:] = q0; d.qvel[:] = 0; d.ctrl[:] = ctrl
    o.set_state(q0)
    for k in range(40):
        mujoco.mj_step(m, d)
        o.step(ctrl)
        if d.ncon == 0 and o.s.ncon == 0:
            np.testing.assert_allclose(o.view("qpos"), d.qpos, rtol=0, atol=1e-9, err_msg=f"substep {k}")
            np.testing.assert_allclose(o.view("qvel"), d.qvel, rtol=0, atol=1e-7, err_msg=f"substep {k}")


def test_config1_against_real_mujoco_reports_first_divergence():
    from oracle import ko_py as ko
    mujoco, m, d = _load()
    q0, hq = scenarios.config1_state("CubeS")
    acts = scenarios.config_actions(1, 30, base_seed=0)[:, :, 0]
    o = ko.OracleSim(ko.OracleModel(scenarios.model_blob("CubeS")), hq, solver_iterations=50)
    o.env_reset(q0)
    d.qpos[:] = q0; d.qvel[:] = 0
    mujoco.mj_forward(m, d)
    first = None
    for t in range(30):
        ctrl = ko.env_ctrl(o.view("geom_xpos").reshape(-1, 3)[1], o.view("geom_xmat").reshape(-1, 9)[1], acts[t].astype(np.float64))[2]
        d.ctrl[:] = ctrl
        for k in range(15):
            mujoco.mj_step(m, d)
            o.step(ctrl)
            rel = np.abs(o.view("qpos") - d.qpos).max() / max(1e-3, np.abs(d.qpos).max())
            if first is None and rel > 1e-4:
                first = (15 * t + k, rel, int(d.ncon), int(o.s.ncon))
            if 15 * t + k < 20:
                assert rel <= 1e-4, (15 * t + k, rel)
    print("oracle vs mujoco", mujoco.__version__, ": first substep beyond 1e-4 relative qpos error (substep, rel, ncon mujoco, ncon oracle):", first)
