"""CPU coverage of the MULTI-GEOM objects (the reference's Bottle / TBottle / Bowl / RBowl models: `object` plus jointless child bodies,
kinova_description/j2s7s300_end_effector_v1_sbottle.xml:158-186 and siblings): the compiled assets, the oracle on them, and the kernel
SOURCE compiled for the host with the multi-geom capacities (-DKS_MULTI_GEOM, what libkinova_sim_mg.so is built with) against the
oracle.  The compiled gfx950 kernels are checked in tests/test_gpu_multi_geom.py (-m gpu)."""
import numpy as np
import pytest

from oracle import ko_py as ko
from kinovagrasping_amd import model_compiler as mc, scenarios
from kinovagrasping_amd.sim import SOLVER_ITERATIONS
from tests.native_build import Lane

PIECES = {"Bottle": 5, "TBottle": 5, "Bowl": 9, "RBowl": 5, "Hour": 3}


def kind(shape):
    return shape[:-1]


@pytest.mark.parametrize("shape", scenarios.MULTI_GEOM_SHAPES)
def test_multi_geom_asset_structure(shape, assets_dir):
    M = mc.read_blob(assets_dir / f"{shape}.ksm")
    k = PIECES[kind(shape)]
    ng = 8 + k
    assert len(M["geom_body"]) == ng and (M["geom_body"][8:] == 9).all() and list(M["geom_mesh"][8:]) == list(range(3, 3 + k))
    # the 30 pairs of every model, then ground + seven hand geoms against every welded piece (dynamic: geom margin, friction 1);
    # the pieces are never paired with each other or with `object`
    P = M["pairs"]
    assert P.shape == (30 + 8 * (k - 1), 5)
    assert (P[:8, 1] == 8).all() and (P[:8, 4] == 0).all() and (P[8:, 4] == 0.001).all() and (P[30:, 2] == 1.0).all()
    extra = {(int(a), int(b)) for a, b in P[30:, :2]}
    assert extra == {(a, b) for b in range(9, ng) for a in range(8)}
    assert not any(a >= 8 for a, b in P[:, :2])
    # composite inertial of the welded pieces
    pm = M["piece_mass"]
    assert len(pm) == k and abs(pm.sum() - M["body_mass"][9]) < 1e-15
    com = (pm[:, None] * M["geom_pos"][8:]).sum(0) / pm.sum()
    assert np.abs(com - M["body_ipos"][9]).max() < 1e-15
    # ... recomputed independently from the pieces' ray triangles (float32 copies of the mesh triangles in the geom frame): legacy mesh
    # inertia per piece at its mass, rotated into the object frame, parallel axes to the composite centre of mass
    Ic = np.zeros((3, 3))
    for j in range(k):
        tri = M[f"mesh{3 + j}_tri"].astype(np.float64).reshape(-1, 3, 3)
        vol, c, I = mc.mesh_mass_properties_legacy(tri)
        assert np.abs(c).max() < 2e-6                                   # the geom frame is centred on the piece's centre of mass
        R = mc.quat_to_mat(M["geom_quat"][8 + j])
        r = M["geom_pos"][8 + j] - com
        Ic += R @ (I * pm[j] / vol) @ R.T + pm[j] * (r @ r * np.eye(3) - np.outer(r, r))
    ev = np.sort(np.linalg.eigvalsh(Ic))
    assert np.allclose(ev, np.sort(M["body_inertia"][9]), rtol=2e-4), (ev, M["body_inertia"][9])
    Ri = mc.quat_to_mat(M["body_iquat"][9])
    assert np.allclose(Ri @ np.diag(M["body_inertia"][9]) @ Ri.T, Ic, rtol=0, atol=2e-4 * ev.max())
    # per-geom inverse weights: MuJoCo keeps every piece as a body; its translational inverse weight is taken at ITS centre of mass
    Minv = np.linalg.inv(M["M0"])[9:, 9:]
    for g in range(8, ng):
        r = M["geom_pos"][g]                                           # qpos0: object frame = world frame
        rx = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
        J = np.concatenate([np.eye(3), -rx], axis=1)                    # d(point velocity) / d(free-joint velocity), body-frame rotation = identity at qpos0
        assert abs(np.trace(J @ Minv @ J.T) / 3 - M["geom_invweight0"][g]) < 1e-9
    assert (M["geom_invweight0"][:8] == M["body_invweight0"][M["geom_body"][:8], 0]).all()
    assert M["geom_invweight0"][8:].min() >= 1.0 / (pm.sum() + 0.01) - 1e-9


def test_object_size_of_multi_geom_objects_follows_the_reference_rule(assets_dir):
    """_get_obj_size (kinova_gripper_env.py:706-746): bowls are constants scaled by the env's size letter, bottles the widest piece and the
    SUM of the pieces' heights; the observation doubles the last entry (:529)"""
    # the bowls' constants are scaled by the env's size letter - which the object schedule never updates: 'm' (0.85) for every bowl on the drivers' path
    for z in "SMB":
        assert np.allclose(mc.read_blob(assets_dir / f"Bowl{z}.ksm")["obj_size_obs"], [0.175 * 0.85, 0.175 * 0.85, 2 * 0.07 * 0.85])
        assert np.allclose(mc.read_blob(assets_dir / f"RBowl{z}.ksm")["obj_size_obs"], [0.17 * 0.85, 0.17 * 0.85, 2 * 0.075 * 0.85])
    # bottles: per piece (walked from the last geom back to `object`) the extents are reordered so that the two most similar ones come
    # first - swap of the first and last unless the first two already are the closest pair -, then widths by maximum, heights summed
    for shape in ("BottleS", "TBottleB"):
        M = mc.read_blob(assets_dir / f"{shape}.ksm")
        w0 = w1 = h = 0.0
        for sz in M["geom_size"][8:][::-1]:
            a, b, c = sz
            if min(abs(b - c), abs(a - c)) < abs(a - b):
                a, c = c, a
            w0, w1, h = max(w0, a), max(w1, b), h + c
        assert np.allclose(M["obj_size_obs"], [w0, w1, 2 * h], rtol=0, atol=1e-15)
    assert np.allclose(mc.read_blob(assets_dir / "BottleS.ksm")["obj_size_obs"], [0.02190571, 0.02187536, 0.16126273], atol=5e-9)   # known answer
    assert 0.25 < mc.read_blob(assets_dir / "TBottleB.ksm")["obj_size_obs"][2] < 0.31


def test_hull_graphs_of_the_pieces_have_no_local_maxima(assets_dir):
    """the kernels find support vertices by hill climbing on the hull graph (ks_core.h hull_climb); the oracle scans.  They agree when
    every vertex that is not the maximiser of a direction has a strictly better neighbour - checked on 300 directions per hull
    (with the support skew of ko_physics.c / ks_core.h applied, as both do)"""
    rng = np.random.default_rng(0)
    for shape in ("BottleS", "TBottleS", "BowlS", "RBowlS", "LemonS", "HourS", "VaseB"):
        M = mc.read_blob(assets_dir / f"{shape}.ksm")
        for s in range(3, len(M["geom_body"]) - 5):
            V, off, adj = M[f"mesh{s}_vert"], M[f"mesh{s}_adj_off"], M[f"mesh{s}_adj"]
            D = rng.normal(size=(300, 3))
            D[:6] = np.concatenate([np.eye(3), -np.eye(3)])             # the axis directions: where flat caps tie
            D = D + 1e-6 * np.abs(D).sum(1, keepdims=True) * np.array([0.5377, -0.6240, 0.5671])
            vals = V @ D.T                                              # [nv, 300]
            nbr_best = np.maximum.reduceat(vals[adj], off[:-1], axis=0)
            stuck = (nbr_best <= vals) & (vals < vals.max(0, keepdims=True))
            assert not stuck.any(), (shape, s, int(stuck.sum()))


def in_hand_start(shape):
    M = mc.read_blob(scenarios.model_blob(shape))
    q = np.zeros(16)
    q[12] = 1.0
    q[9:12] = -M["geom_pos"][8] * np.array([1.0, 1.0, 0.0])
    return q


@pytest.mark.parametrize("shape,pose", [("BottleS", "normal"), ("TBottleM", "normal"), ("BowlS", "normal"), ("RBowlB", "normal"),
                                        ("BottleS", "top"), ("TBottleS", "top"), ("BowlS", "rotated"), ("RBowlS", "rotated"),
                                        ("HourM", "normal"), ("LemonM", "normal"), ("LemonS", "top")])     # (Lemon: ONE geom, but a 2434-vertex hull: the multi-geom library's tables in global memory)
def test_multi_geom_kernel_source_reproduces_oracle_substeps(shape, pose):
    """the kernel source with the multi-geom capacities, one lane on the host, against the oracle through a scripted grasp that touches
    welded pieces (plane contacts of the pieces, finger- and palm-piece hull pairs, lift) in the three hand poses: fp64 to round-off, fp32
    within its one-step bounds"""
    blob = scenarios.model_blob(shape)
    m = ko.OracleModel(blob)
    hq = scenarios.hand_quat_for(pose)
    lane64, lane32 = Lane(blob, 64, multi_geom=True), Lane(blob, 32, multi_geom=True)
    s = ko.OracleSim(m, hq, solver_iterations=SOLVER_ITERATIONS)
    s.s.rays_enabled = 0
    q0 = in_hand_start(shape)
    q0[0:3] = scenarios.hand_slide_offsets(pose, shape, "pose")
    s.set_state(q0)
    s.forward()
    ctrl = np.zeros(9); ctrl[6:9] = 0.6
    e64, e32, pairs = [], [], set()
    for i in range(320):
        if i == 200:
            ctrl[4] = 0.4
        before = (s.view("qpos").copy(), s.view("qvel").copy(), s.view("qacc_warmstart").copy())
        s.step(ctrl)
        pairs |= {(c["geom1"], c["geom2"]) for c in s.contacts()}
        qp, qv, qw, nc, con, st = lane64.substep(*before, ctrl, hq)
        assert nc == s.s.ncon and st == 0
        e64.append(max(np.abs(qp - s.view("qpos")).max(), 1e-2 * np.abs(qv - s.view("qvel")).max()))
        qp, qv, qw, nc, con, st = lane32.substep(*before, ctrl, hq)
        assert st == 0
        e32.append(np.abs(qp - s.view("qpos")).max())
    print(f"{shape} {pose}: pairs {sorted(pairs)}; fp64 lane worst {max(e64):.2e}; fp32 lane median {np.median(e32):.2e} max {max(e32):.2e}")
    assert (any(b > 8 for a, b in pairs) or shape.startswith("Lemon")) and any(a > 0 and b >= 8 for a, b in pairs)
    assert max(e64) < 1e-9
    assert np.median(e32) < 2e-7 and max(e32) < 3e-3


def test_env_step_and_rays_of_a_multi_geom_object_kernel_source_vs_oracle():
    """whole env.step()s: 15 substeps, the 17 rangefinders over every piece's triangles, observation, reward (fp64 lane = oracle)"""
    shape = "BottleS"
    blob = scenarios.model_blob(shape)
    hq = scenarios.hand_quat_for("normal")
    o = ko.OracleSim(ko.OracleModel(blob), hq, solver_iterations=SOLVER_ITERATIONS)
    lane = Lane(blob, 64, multi_geom=True)
    q0 = in_hand_start(shape)
    ref = o.env_reset(q0)
    obs, rays = lane.reset_obs(q0, hq)
    assert np.abs(obs - ref).max() < 1e-12
    assert (rays > 0).sum() >= 3                                       # the fingers' rangefinders see the bottle's pieces
    st = (q0, np.zeros(15), np.zeros(15))
    for t in range(4):
        a = np.array([0.0, 0.6, 0.6, 0.6])
        ro, rr, rd, _ = o.env_step(a)
        qp, qv, qw, ob, rew, done, rays, status = lane.env_step(*st, hq, a)
        st = (qp, qv, qw)
        assert status == 0 and np.abs(ob - ro).max() < 1e-9 and rew == rr and done == rd


def test_standard_build_refuses_a_multi_geom_blob(capfd):
    with pytest.raises(AssertionError):
        Lane(scenarios.model_blob("BowlS"), 64, multi_geom=False)
    assert "multi-geom" in capfd.readouterr().err
    with pytest.raises(AssertionError):                              # ... and a single-geom object whose hull is beyond its 1024 vertices
        Lane(scenarios.model_blob("LemonS"), 64, multi_geom=False)
    assert "1 .. 1024 vertices" in capfd.readouterr().err
    from kinovagrasping_amd import sim as ks
    assert ks.blob_needs_mg_library(scenarios.model_blob("LemonB")) and not ks.blob_is_multi_geom(scenarios.model_blob("LemonB"))
    assert ks.blob_is_multi_geom(scenarios.model_blob("HourM")) and not ks.blob_needs_mg_library(scenarios.model_blob("VaseS"))


def test_multi_geom_library_exports_the_simulator_abi():
    import re
    from pathlib import Path
    from kinovagrasping_amd import build as kb, sim as ks
    kb.build()
    L = ks.load_library(multi_geom=True)
    header = re.sub(r"/\*.*?\*/", "", (Path(__file__).resolve().parents[1] / "include" / "kinova_sim.h").read_text(), flags=re.S)
    for name in sorted(set(re.findall(r"\b(ks_[a-z_0-9]+)\s*\(", header))):
        assert hasattr(L, name), f"{name} declared in include/kinova_sim.h but not exported by libkinova_sim_mg.so"
    assert ks.blob_is_multi_geom(scenarios.model_blob("RBowlM")) and not ks.blob_is_multi_geom(scenarios.model_blob("CubeS"))


def test_start_tables_and_the_fallback_rule():
    """start rows exist for what the reference ships (Bottle / TBottle: all three classes; Bowl / RBowl: rotated and top); elsewhere the
    reference's empty-file rule (randomize_initial_pos_data_collection, kinova_gripper_env.py:821-849) draws the start"""
    assert scenarios.has_start_table("BottleS", "normal") and scenarios.start_coord_table("TBottleB", "top").shape[1] == 3
    assert scenarios.has_start_table("BowlM", "rotated") and not scenarios.has_start_table("BowlS", "normal") and not scenarios.has_start_table("RBowlS", "normal")
    rng = np.random.RandomState(0)
    p = np.array([scenarios.fallback_start("BowlS", "normal", rng) for _ in range(200)])
    assert np.hypot(p[:, 0], p[:, 1]).max() <= 0.14875 / 2 and np.allclose(p[:, 2], 0.0595 / 2)
    assert np.allclose(scenarios.fallback_start("BowlS", "rotated", rng), [0, 0, 0.02975])


def test_reset_correction_of_the_reference_moves_the_object_geom_onto_the_commanded_point():
    """KinovaGripper_Env.reset (kinova_gripper_env.py:1367-1386): commanded point -> free joint, read the `object` geom's pose back, and if it is
    more than 5 cm off write commanded + (commanded - pose).  Restated literally on the oracle and compared with scenarios.reset_body_position."""
    cmd = np.array([0.0425, 0.004, -0.01])
    for shape, moved in (("CubeS", False), ("Vase2B", False), ("BottleS", True), ("BowlM", True), ("HourS", True), ("LemonS", True), ("RBowlB", True)):
        o = ko.OracleSim(ko.OracleModel(scenarios.model_blob(shape)), scenarios.hand_quat_for("normal"), solver_iterations=SOLVER_ITERATIONS)
        q = np.zeros(16); q[12] = 1.0; q[9:12] = cmd
        o.set_state(q); o.forward()
        pose = o.view("geom_xpos").reshape(-1, 3)[8].copy()
        deltas = cmd - pose
        expect = cmd + deltas if np.linalg.norm(deltas) > 0.05 else cmd
        got = scenarios.reset_body_position(shape, cmd)
        assert np.abs(got - expect).max() < 1e-12 and (np.abs(got - cmd).max() > 0.01) == moved, shape
        q[9:12] = got
        o.set_state(q); o.forward()
        if moved:
            assert np.abs(o.view("geom_xpos").reshape(-1, 3)[8] - cmd).max() < 1e-12
