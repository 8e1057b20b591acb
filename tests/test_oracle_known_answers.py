"""Known answers for the physics oracle (oracle/ko_physics.c) that do NOT come from its author's reading of MuJoCo's
source: closed-form mechanics of the model (XML:7-291 of the reference's j2s7s300_end_effector_v1_CubeS.xml, constants
through the compiled blob) and the mathematics of the constraint problem itself (primal / dual optimality, two solvers
that share no code path).  CPU only.  The oracle's physics stays 'parity unpinned' against real MuJoCo 1.50 (no binary,
no golden vectors in the reference: DESIGN.md section 2) - these tests bound what could be wrong with it."""
import numpy as np
import pytest

from oracle import ko_py as ko
from kinovagrasping_amd import scenarios

G = 9.81
H = 0.01                       # timestep, XML:9
DAMP, ARM = 0.2, 0.01          # default joint damping / armature on every dof, XML:41
M_OBJ = 0.1                    # XML:153
M_HAND = 0.727 + 6 * 0.01      # link_7 + six finger links, XML:60,81,90


@pytest.fixture(scope="module")
def model():
    return ko.OracleModel(scenarios.model_blob("CubeS"))


def fresh(model, qobj=(0.0, -0.3, 3.0), iters=20):
    s = ko.OracleSim(model, scenarios.hand_quat_for("normal"), solver_iterations=iters)
    q0 = np.zeros(16); q0[9:12] = qobj; q0[12] = 1
    s.set_state(q0)
    return s


def test_free_fall_with_armature_and_implicit_damping(model):
    """Object in free flight (no contact, the hand far away): semi-implicit Euler with the joint damping treated
    implicitly, per translational dof  (m + armature + h d) a' = f - d v,  v += h a',  z += h v  (closed-form recurrence;
    armature adds inertia but no weight)."""
    s = fresh(model)
    ctrl = np.zeros(9); ctrl[5] = 0.2932
    vz = vx = 0.0
    s.view("qvel")[9] = vx = 0.3                     # a horizontal throw: pure damping on x
    z, x = 3.0, 0.0
    for k in range(60):
        s.step(ctrl)
        az = (-M_OBJ * G - DAMP * vz) / (M_OBJ + ARM + H * DAMP)
        ax = (-DAMP * vx) / (M_OBJ + ARM + H * DAMP)
        vz += H * az; vx += H * ax
        z += H * vz; x += H * vx
        assert not any(8 in (c["geom1"], c["geom2"]) for c in s.contacts())           # nothing touches the object
        np.testing.assert_allclose([s.view("qvel")[11], s.view("qvel")[9]], [vz, vx], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose([s.view("qpos")[11], s.view("qpos")[9]], [z, x], rtol=1e-12, atol=1e-15)
    # rotation: a free spin about a principal axis decays by the damping alone: (I + armature + h d) w' = -d w
    s = fresh(model)
    w = 2.0
    s.view("qvel")[14] = w
    Izz = _principal_inertia(model)[2]
    for k in range(20):
        s.step(ctrl)
        w += H * (-DAMP * w) / (Izz + ARM + H * DAMP)
        np.testing.assert_allclose(s.view("qvel")[14], w, rtol=1e-10)


def _principal_inertia(model):
    from kinovagrasping_amd import model_compiler as mc
    return mc.read_blob(scenarios.model_blob("CubeS"))["body_inertia"][9]


def test_resting_cube_carries_its_weight(model):
    """At rest on the plane the contact normal forces sum to m g whatever the soft-constraint parameters are (Newton's
    second law at qacc = 0), the friction forces vanish, and the cube rests ~1 um INSIDE the floor: the explicit object-ground pair
    has margin 0 (MuJoCo's pair default), so its contacts only exist on penetration."""
    s = fresh(model, qobj=(0.05, 0.0, 0.0654))
    ctrl = np.zeros(9); ctrl[5] = 0.2932
    for k in range(400):
        s.step(ctrl)
    f = s.contact_forces()
    ground = [i for i, c in enumerate(s.contacts()) if c["geom1"] == 0 and c["geom2"] == 8]
    assert len(ground) == 4
    assert abs(f[ground, 0].sum() - M_OBJ * G) < 1e-6 * M_OBJ * G
    assert np.abs(f[ground, 1:]).max() < 1e-7
    assert np.abs(s.view("qvel")[9:15]).max() < 1e-7
    d = np.array([s.contacts()[i]["dist"] for i in ground])
    assert (d < 0).all() and (d > -5e-6).all()


def test_sliding_cube_decelerates_by_coulomb_friction(model):
    """A cube thrown along the ground (mu = 0.3, XML:159) slides along tangent 1 of its contacts' frames (world y for a
    normal along z).  While it slides in contact exactly one edge of the +-t1 pair of the friction PYRAMID is loaded, so
    |F_t| = mu * N to round-off, against the motion, nothing along t2; the loaded edge also carries normal force (the
    pyramid's well-known coupling: the cube hops), and it comes to rest in about v0 (m + armature) / (mu m g)."""
    s = fresh(model, qobj=(0.3, -0.2, 0.0489))
    ctrl = np.zeros(9); ctrl[5] = 0.2932
    for k in range(50):
        s.step(ctrl)                                              # settle
    v0 = 0.4
    s.view("qvel")[10] = v0
    sliding, steps = 0, 0
    while steps < 400 and (steps < 5 or abs(s.view("qvel")[10]) > 1e-3):
        s.step(ctrl); steps += 1
        f = s.contact_forces()
        ground = [i for i, c in enumerate(s.contacts()) if c["geom1"] == 0 and c["geom2"] == 8]
        N, Ft1, Ft2 = f[ground, 0].sum(), f[ground, 1].sum(), f[ground, 2].sum()
        if N > 0 and s.view("qvel")[10] > 0.02:
            sliding += 1
            assert Ft1 < 0 and abs(abs(Ft1) - 0.3 * N) <= 1e-9 * N and abs(Ft2) <= 1e-9 * N, (steps, N, Ft1, Ft2)
    assert sliding >= 3
    t_stop = v0 * (M_OBJ + ARM) / (0.3 * M_OBJ * G)
    assert 0.5 * t_stop < steps * H < 1.5 * t_stop, (steps * H, t_stop)
    for k in range(60):
        s.step(ctrl)
    f = s.contact_forces()
    ground = [i for i, c in enumerate(s.contacts()) if c["geom1"] == 0 and c["geom2"] == 8]
    assert abs(f[ground, 0].sum() - M_OBJ * G) < 1e-3 * M_OBJ * G and np.abs(f[ground, 1:]).max() < 1e-4     # at rest again


def test_hover_sag_and_tendon_coupling(model):
    """Zero action: the feed-forward motor (gear 25 x ctrl 0.2932 = 7.33 N, ENV:1511-1515) leaves (m_hand g - 7.33) N to
    the velocity servo (kv 150) and the joint damping: steady sag v = -(m g - 7.33) / (150 + 0.2).
    Closing the fingers: the fixed tendon q_prox - 2 q_dist = 0 (XML:171-188) keeps the distal joint at half the
    proximal angle while the servo (kv 2.5) drives it at the commanded rate."""
    s = fresh(model)
    ctrl = np.zeros(9); ctrl[5] = 0.2932
    for k in range(300):
        s.step(ctrl)
    v_pred = -(M_HAND * G - 25 * 0.2932) / (150 + DAMP)
    np.testing.assert_allclose(s.view("qvel")[2], v_pred, rtol=2e-3)         # the slide along world z in the 'normal' pose
    assert abs(v_pred + 2.6e-3) < 1e-4
    ctrl[6:9] = 0.5
    for k in range(150):
        s.step(ctrl)
    q = s.view("qpos")
    for f in range(3):
        assert 0.5 < q[3 + 2 * f] < 0.8
        assert abs(q[4 + 2 * f] - q[3 + 2 * f] / 2) < 0.02 * q[3 + 2 * f]
    np.testing.assert_allclose(s.view("qvel")[3:9:2], 0.5, rtol=0.1)


def _constraint_problem(s):
    n = s.s.nefc
    J = s.view("efc_J").reshape(-1, 15)[:n].copy()
    R, aref, typ = s.view("efc_R")[:n].copy(), s.view("efc_aref")[:n].copy(), s.view("efc_type")[:n].copy()
    M, a_s = s.view("M").reshape(15, 15).copy(), s.view("qacc_smooth").copy()
    return J, R, aref, typ, M, a_s


def test_newton_and_pgs_meet_at_the_same_optimum(model):
    """The constraint solver from two sides.  Newton minimises the PRIMAL cost over accelerations; projected Gauss-Seidel
    (the solver BASELINE's north_star names) descends the DUAL cost over forces - different variables, different
    algorithms, no shared code.  At the optimum of this strictly convex problem primal cost = - dual cost (strong
    duality) and qacc = qacc_smooth + M^-1 J^T f.  Also: the oracle's PGS is checked against an independent numpy
    PGS sweep for sweep (identical), its dual cost falls monotonically, and the max-norm acceleration error is NOT
    monotone along the way - which is why round 1's table (profiles/r01_pgs_vs_newton.txt: 4.2e-3 at 1000 sweeps,
    1.2e-2 at 2000) looked wrong and is not."""
    s = fresh(model, qobj=(0.052897, 0.000732, 0.0654), iters=30)
    ctrl = np.zeros(9); ctrl[5] = 0.2932; ctrl[6:9] = 0.3
    states = {}
    for i in range(200):
        s.step(ctrl)
        if i in (7, 100, 199):
            states[i] = (s.view("qpos").copy(), s.view("qvel").copy(), s.view("qacc_warmstart").copy())
    for i, (qp, qv, qw) in states.items():
        s.s.solver, s.s.solver_iterations = 0, 60
        s.set_state(qp, qv, qw); s.view("ctrl")[:] = ctrl; s.forward()
        a_newton, f_newton = s.view("qacc").copy(), s.view("efc_force")[:s.s.nefc].copy()
        J, R, aref, typ, M, a_s = _constraint_problem(s)
        Minv = np.linalg.inv(M)
        A, b = J @ Minv @ J.T + np.diag(R), J @ a_s - aref
        dual = lambda f: 0.5 * f @ A @ f + f @ b
        jar = J @ a_newton - aref
        active = (typ == 0) | (jar < 0)
        primal = 0.5 * (a_newton - a_s) @ M @ (a_newton - a_s) + 0.5 * (jar[active] ** 2 / R[active]).sum()
        # Newton's point satisfies the KKT conditions of the dual: forces from the primal map, feasibility, complementarity
        np.testing.assert_allclose(f_newton, np.where(active, -jar / R, 0.0), rtol=1e-9, atol=1e-12)
        assert (f_newton[typ != 0] >= 0).all()
        np.testing.assert_allclose(a_s + Minv @ J.T @ f_newton, a_newton, rtol=1e-9, atol=1e-9)
        assert abs(primal + dual(f_newton)) <= 1e-9 * max(1.0, abs(primal))            # strong duality at Newton's point
        # PGS: oracle == numpy, monotone dual cost, converging to Newton's optimum
        f, costs, errs = np.zeros(len(b)), [], []
        # the oracle's warm start (forces implied by qacc_warmstart, kept if they beat zero) reproduced
        fw = -(J @ qw - aref) / R
        fw[(typ != 0) & (fw < 0)] = 0
        if dual(fw) <= 0:
            f = fw.copy()
        done = 0
        for sweeps in (10, 100, 1000, 4000):
            for _ in range(sweeps - done):
                for r in range(len(b)):
                    fi = f[r] - (b[r] + A[r] @ f) / A[r, r]
                    f[r] = fi if (typ[r] == 0 or fi > 0) else 0.0
            done = sweeps
            s.s.solver, s.s.solver_iterations = 1, sweeps
            s.set_state(qp, qv, qw); s.forward()
            np.testing.assert_allclose(s.view("efc_force")[:len(b)], f, rtol=1e-9, atol=1e-12)
            costs.append(dual(f)); errs.append(np.abs(s.view("qacc") - a_newton).max())
        assert all(c1 <= c0 + 1e-12 for c0, c1 in zip(costs, costs[1:]))
        assert costs[-1] >= dual(f_newton) - 1e-9 and costs[-1] - dual(f_newton) < 1e-4 * abs(dual(f_newton))
        assert errs[-1] < 0.02 * max(1.0, np.abs(a_newton).max())


def test_newton_cap_of_the_product_setting_is_never_the_exit(model):
    """kernel and oracle shared a cap of 6 Newton iterations, so the parity tests could not see solver
    truncation.  tests/studies/solver_cap.py (profiles/r03_solver_cap.txt): at cap 6, 2.3 % of 13 200 grasp / lift substeps end
    at the cap with up to 20 % error in qacc; no problem needs more than 10.  Here, on one closing-grasp trajectory: the product
    setting SOLVER_ITERATIONS converges by its stop rule in every substep and equals the cap-100 / 1e-10 answer, cap 6 does not."""
    from kinovagrasping_amd.sim import SOLVER_ITERATIONS
    from tests.studies.solver_cap import scenario
    _, hi = scenario(("CubeS", "normal", 100, SOLVER_ITERATIONS))
    assert hi["capped"].sum() == 0 and hi["dacc"].max() < 1e-9 and hi["drift"].max() < 1e-9 and hi["iters_a"].max() <= 12
    _, lo = scenario(("CubeS", "normal", 100, 6))
    assert lo["capped"].sum() > 0 and lo["dacc"].max() > 1e-2


def test_rangefinder_sees_only_the_front_face_of_the_ground_plane(model):
    """mju_rayGeom, plane case: a ray whose direction does not point at the +z (front) side of the plane is rejected.  Matters in
    the 'rotated' / 'top' fresh-env starts whose hand is partly below z = 0.  Palm sites look along the hand's
    -z (site quat 0 1 0 0): hand upright and 0.255 m up -> the palm-centre ray reads its height; hand flipped and pushed 0.125 m
    under the floor, looking UP through it -> no hit (rounds 1-2 returned 0.125)."""
    far = np.zeros(16); far[9:12] = [0.5, -0.5, 0.0479]; far[12] = 1
    up = ko.OracleSim(model, np.array([1.0, 0.0, 0.0, 0.0]))
    q = far.copy(); q[1] = 0.3
    up.set_state(q); up.forward()
    z = up.view("site_xpos").reshape(17, 3)[0, 2]
    assert abs(z - 0.2554) < 1e-3 and abs(up.view("sensordata")[9] - z) < 1e-12
    down = ko.OracleSim(model, np.array([0.0, 1.0, 0.0, 0.0]))
    down.set_state(q); down.forward()
    assert down.view("site_xpos").reshape(17, 3)[0, 2] < -0.12 and down.view("site_xmat").reshape(17, 9)[0, 8] > 0.999
    assert (down.view("sensordata")[9:14] == -1).all()
