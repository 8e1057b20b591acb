"""GPU parity tests: the HIP path (through the C ABI, libkinova_sim.so) against the fp64 CPU oracle on
the same seeded inputs.  Run with `-m gpu` on an MI355X.

Tolerances (fp32 product vs fp64 oracle), stated once:
  * one mj_step from identical states (teacher forced): |dqpos|_inf <= 5e-7 for >= 95% of the states
    and median <= 1e-7; the tail (<= 3% of the states above 1e-5) is the single-point contact *position* on
    parallel features (DESIGN.md "known limits"), for which the bound is 3e-3.
  * free-running config-1 episode: relative qpos error <= 1e-4 (north_star) for at least the first
    200 substeps; afterwards contact make/break makes trajectories chaotic (SURVEY hard part 2) and
    the test only reports the first substep that exceeds the tolerance.
  * fp64 instantiation of the same kernels: <= 1e-9 everywhere (algorithm exactness).
"""
import numpy as np
import pytest
import torch

from oracle import ko_py as ko
from kinovagrasping_amd import scenarios
from kinovagrasping_amd.sim import SOLVER_ITERATIONS

pytestmark = pytest.mark.gpu


def _sim(*a, **k):
    from kinovagrasping_amd.sim import KinovaSim
    return KinovaSim(*a, **k)


@pytest.fixture(scope="module")
def cube(assets_dir):
    return ko.OracleModel(scenarios.model_blob("CubeS"))


def oracle_grasp_trajectory(model, n_sub=330, x0=0.0, y0=0.0, iters=SOLVER_ITERATIONS):
    """closing fingers, then lifting: returns per-substep (state before, ctrl, state after)"""
    hq = scenarios.hand_quat_for("normal")
    s = ko.OracleSim(model, hq, solver_iterations=iters)
    q0 = np.zeros(16); q0[9:12] = [x0, y0, 0.0654]; q0[12] = 1
    s.env_reset(q0)
    ctrl = np.zeros(9); ctrl[5] = 0.2932; ctrl[6:9] = 0.5
    rec = []
    for i in range(n_sub):
        if i == 250:
            ctrl[4] = 0.5
        before = (s.view("qpos").copy(), s.view("qvel").copy(), s.view("qacc_warmstart").copy())
        s.step(ctrl)
        after = (s.view("qpos").copy(), s.view("qvel").copy(), s.view("qacc_warmstart").copy())
        rec.append((before, ctrl.copy(), after, s.s.ncon))
    return hq, rec


def run_teacher_forced(precision, model, rec, hq, iters=SOLVER_ITERATIONS, shape="CubeS"):
    n = len(rec)
    sim = _sim(n, shape, precision=precision, solver_iterations=iters)
    dt = sim.dtype
    q0 = np.stack([r[0][0] for r in rec], 1)
    sim.reset(torch.as_tensor(q0), torch.as_tensor(np.repeat(hq[:, None], n, 1)))
    sim.set_state(torch.as_tensor(q0), torch.as_tensor(np.stack([r[0][1] for r in rec], 1)), torch.as_tensor(np.stack([r[0][2] for r in rec], 1)))
    sim.substep(torch.as_tensor(np.stack([r[1] for r in rec], 1)))
    st = sim.get_state()
    torch.cuda.synchronize()
    qp = st["qpos"].double().cpu().numpy()
    qv = st["qvel"].double().cpu().numpy()
    ncon = st["ncon"].cpu().numpy()
    eq = np.abs(qp - np.stack([r[2][0] for r in rec], 1)).max(0)
    ev = np.abs(qv - np.stack([r[2][1] for r in rec], 1)).max(0)
    onc = np.array([r[3] for r in rec])
    assert (st["status"].cpu().numpy() == 0).all()
    sim.close()
    return eq, ev, ncon, onc


def test_one_step_fp64_matches_oracle(cube):
    hq, rec = oracle_grasp_trajectory(cube)
    eq, ev, ncon, onc = run_teacher_forced(64, cube, rec, hq)
    print(f"fp64 one-step |dqpos|: median {np.median(eq):.2e} p99 {np.percentile(eq, 99):.2e} max {eq.max():.2e}")
    assert (ncon == onc).all()
    # every state, no percentile: kernel and oracle share the closest-point formulation (closest_tri), so even the
    # single-point contact positions on parallel features break their ties alike
    assert eq.max() < 1e-9, eq.max()
    assert ev.max() < 1e-7, ev.max()


def test_one_step_fp32_matches_oracle(cube):
    hq, rec = oracle_grasp_trajectory(cube)
    eq, ev, ncon, onc = run_teacher_forced(32, cube, rec, hq)
    print(f"one-step |dqpos|: median {np.median(eq):.2e} p95 {np.percentile(eq, 95):.2e} max {eq.max():.2e};"
          f" contact-count mismatches {int((ncon != onc).sum())}/{len(onc)}")
    tail = eq > 1e-5
    print(f" states above 1e-5: {int(tail.sum())}/{len(eq)}, p99 {np.percentile(eq, 99):.2e}")
    # measured at the end of round 5 (fp64 read-offs of MPR's final portal and of the plane pairs' vertex distances, float32 hull tables in the model):
    # median 3.2e-8, p95 9.3e-8, max 5.0e-7, no state above 1e-5, no contact-count mismatch (round 4: p95 <= 5e-7, 3 % above 1e-5, max 3e-3 allowed)
    assert np.median(eq) <= 1e-7
    assert np.percentile(eq, 95) <= 2.5e-7
    # thresholds = measured x 3 (round 6 measures the same 3.2e-8 / 9.3e-8 / 5.0e-7): no state above 1e-5 any more
    assert tail.sum() == 0
    assert eq.max() <= 1.5e-6
    assert (ncon != onc).sum() == 0


@pytest.mark.parametrize("shape", ["mbox", "bbox", "scyl", "mcyl", "bcyl"])
def test_primitive_objects_one_step_matches_oracle(shape):
    """The env's default model (..._mbox.xml, ENV:62) and its primitive siblings through the HIP kernels: a drop onto the plane,
    the rest on it and a closing grasp, every substep teacher-forced from the oracle's state - fp64 to round-off, fp32 with the
    CubeS bounds (box / cylinder objects are compiled to their convex polytopes, model_compiler.compile_model)."""
    from kinovagrasping_amd import model_compiler as mc
    blob = scenarios.model_blob(shape)
    model = ko.OracleModel(blob)
    hq = scenarios.hand_quat_for("normal")
    o = ko.OracleSim(model, hq, solver_iterations=SOLVER_ITERATIONS)
    half_h = mc.read_blob(blob)["geom_size"][8][2]
    q0 = np.zeros(16); q0[9:12] = [0.0, 0.0, half_h + 0.01]; q0[12] = 1
    o.env_reset(q0)
    ctrl = np.zeros(9); ctrl[5] = 0.2932
    rec, touched = [], False
    for i in range(300):
        if i == 60:
            ctrl[6:9] = 0.6
        before = (o.view("qpos").copy(), o.view("qvel").copy(), o.view("qacc_warmstart").copy())
        o.step(ctrl)
        rec.append((before, ctrl.copy(), (o.view("qpos").copy(), o.view("qvel").copy(), o.view("qacc_warmstart").copy()), o.s.ncon))
        touched = touched or any(c["geom2"] == 8 and c["geom1"] in (2, 3, 4, 5, 6, 7) for c in o.contacts())
    assert touched                                              # the fingers reached the object
    box = shape.endswith("box")
    eq, ev, ncon, onc = run_teacher_forced(64, model, rec, hq, shape=shape)
    print(f"{shape}: fp64 one-step |dqpos| max {eq.max():.2e}, states above 1e-9: {int((eq > 1e-9).sum())}/{len(eq)}")
    assert (ncon == onc).all()
    if box:
        assert eq.max() < 1e-9 and ev.max() < 1e-7, (eq.max(), ev.max())
    else:
        # A cylinder standing on its base has 64 rim vertices at the same depth to the last bit or two: which of them is "the
        # deepest" (and with it the greedy choice of the other three contacts, 0.3 rbound apart) is decided by rounding, and the
        # kernel's fused multiply-adds round differently from the oracle's host arithmetic.  The contact SET then differs (same
        # count, same physics to first order); the states without such a tie are exact.
        assert (eq > 1e-9).mean() <= 0.05 and eq.max() < 1e-3, ((eq > 1e-9).mean(), eq.max())
    eq, ev, ncon, onc = run_teacher_forced(32, model, rec, hq, shape=shape)
    print(f"{shape}: fp32 one-step |dqpos| median {np.median(eq):.2e} p95 {np.percentile(eq, 95):.2e} max {eq.max():.2e}; ncon mismatches {int((ncon != onc).sum())}")
    # flat faces resting on the plane and against the finger pads: more states with parallel features than the cube in a pinch;
    # the standing cylinder's rim tie (above) is broken differently in every fp32 state
    # (measured, end of round 5: median 2.0e-8, p95 9e-8 for all five primitives, max 1.0e-5 (mbox) / <= 1.9e-6 (the others), no contact-count mismatch)
    # (round 6, the penetration query on fp64 points: max 1.0e-6 (mbox), 1.9e-6 (bbox), 1.7e-6 (scyl), <= 2.8e-7 (mcyl, bcyl); threshold = worst x 3)
    assert np.median(eq) <= 6e-8 and np.percentile(eq, 95) <= 3e-7 and eq.max() <= 6e-6 and (ncon != onc).sum() == 0


def test_config1_episode_free_running(cube):
    """BASELINE config 1: one CubeS env, PCG64(0) actions, 30 env-steps, GPU fp32 vs oracle."""
    q0, hq = scenarios.config1_state("CubeS")
    acts = scenarios.config_actions(1, 30, base_seed=0)[:, :, 0]
    o = ko.OracleSim(cube, hq, solver_iterations=SOLVER_ITERATIONS)
    obs_o = [o.env_reset(q0)]
    sim = _sim(1, "CubeS", solver_iterations=SOLVER_ITERATIONS, horizon=0)
    obs_g = [sim.reset(torch.as_tensor(q0[:, None]), torch.as_tensor(hq[:, None])).double().cpu().numpy()[0].copy()]
    first_bad = None
    for t in range(30):
        ob, r, d, info = o.env_step(acts[t])
        og, rg, dg, ig = sim.step(torch.as_tensor(acts[t][:, None]))
        torch.cuda.synchronize()
        qg = sim.get_state()["qpos"].double().cpu().numpy()[:, 0]
        qo = o.view("qpos")
        rel = np.abs(qg - qo).max() / max(1e-3, np.abs(qo).max())
        if first_bad is None and rel > 1e-4:
            first_bad = (t, rel)
        obs_o.append(ob)
        obs_g.append(og.double().cpu().numpy()[0].copy())
        if first_bad is None:
            assert rg.item() == r and bool(dg.item() & 1) == d
    print("config1: first env-step with rel qpos error > 1e-4:", first_bad)
    # reset observation: pure kinematics + rays, must agree tightly
    np.testing.assert_allclose(obs_g[0], obs_o[0], rtol=2e-4, atol=2e-5)
    assert first_bad is None or first_bad[0] * 15 >= 200, first_bad
    sim.close()


def test_batch_config2_first_steps(cube):
    """128 envs of config 2 (different start rows / action streams): 3 env-steps against 128 oracle runs."""
    n = 128
    q0, hq = scenarios.config2_states(n)
    acts = scenarios.config_actions(n, 3)
    sim = _sim(n, "CubeS", solver_iterations=SOLVER_ITERATIONS)
    og = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
    orc = [ko.OracleSim(cube, hq[:, i], solver_iterations=SOLVER_ITERATIONS) for i in range(n)]
    oo = np.stack([orc[i].env_reset(q0[:, i]) for i in range(n)])
    torch.cuda.synchronize()
    np.testing.assert_allclose(og.double().cpu().numpy(), oo, rtol=2e-4, atol=2e-5)
    worst = 0.0
    for t in range(3):
        g = sim.step(torch.as_tensor(acts[t]))[0]
        torch.cuda.synchronize()
        qg = sim.get_state()["qpos"].double().cpu().numpy()
        for i in range(n):
            orc[i].env_step(acts[t][:, i])
        qo = np.stack([orc[i].view("qpos").copy() for i in range(n)], 1)
        rel = np.abs(qg - qo).max(0) / np.maximum(1e-3, np.abs(qo).max(0))
        worst = max(worst, np.median(rel))
        print(f" t={t}: median {np.median(rel):.2e}, share below 1e-4: {(rel < 1e-4).mean():.3f}, max {rel.max():.2e}")
        assert np.median(rel) < 2.5e-7, (t, np.median(rel))
        # tail (measured 0.977 / 0.969 / 0.969 of 128 in the middle of round 5; 1.000 with max 9.4e-7 at its end): a finger link that touches the cube's vertical edge a few um deep - libccd's
        # penetration direction is the direction of the closest point of the final portal, ill-conditioned as the depth approaches MPR's
        # 1e-6 tolerance: same point, same depth, normal 2.6 degrees apart between fp32 and fp64 (env 112, substep 3: depth 5.0e-6)
        # (round 6: 1.000 / max 8.6e-7, 2.5e-7, 6.1e-7 over the three env-steps; threshold = measured x 3)
        assert (rel < 1e-4).mean() >= 0.99 and rel.max() < 3e-6, (t, (rel < 1e-4).mean(), rel.max())
    print("config2 x128: worst median relative qpos error over 3 env-steps", worst)
    sim.close()


# share of each shape's 12 grasp-and-lift envs (3 poses x 4 starts) within 1e-4 relative at substep 200, measured in round 5
# (profiles/r05_long_horizon.txt: 146 of 168 = 87 %; round 4: 99).  What round 5 changed, in order: MuJoCo's operand order in the convex queries (106);
# depth / direction of MPR's final portal read off in fp64 (KS_REFINE_F64: 130); the hulls' geom-frame vertices float32 in the MODEL as in
# mjModel.mesh_vert - the oracle and the fp32 product now hold the same tables - with the plane pairs' vertex distances formed in fp64
# (KS_PLANE_F64 = 2: 146).  Round 6: the penetration query on fp64 Minkowski points (ks_core.h: mpr_penetration_sm): 159 through ks_substep (this test),
# 162 through ks_step (the next one).  One env of slack per shape and two in total (VERDICT r5 next #1a)
LONG_HORIZON_MEASURED = {"CubeS": 12, "CubeB": 11, "CylinderS": 12, "CylinderB": 10, "Cube45S": 12, "Cube45B": 12, "Cone1S": 10, "Cone1B": 11, "Cone2S": 11,
                         "Cone2B": 10, "Vase1S": 11, "Vase1B": 12, "Vase2S": 12, "Vase2B": 11}          # round 6, final build (two-lane penetration query): 157 of 168;
#                          the one-lane build before it 159 with CylinderB 12, Cone1B 9 - the query itself returns the same bits (tools/r06/ab_bits.py, -DKS_SPLIT_CHECK);
#                          through ks_step, the product's path, both builds give 162 (round 5: 146, round 4: 99)
# (the per-shape split moves by an env or two between BUILDS of the same arithmetic - the query out of line gave 157 with Cone1B 11, CylinderB 10 -: the
#  envs on the edge sit at a facet jump of a polygonal "round" surface, and fp contraction of a few expressions differs with inlining.  Re-measure after a
#  change to the stepping kernels; the floors are measured - 1 per shape, - 2 in total, as VERDICT r5 asked)


def test_batched_long_horizon_parity_200_substeps():
    """north_star's bar, batched: 256 BASELINE config-2 envs (random +-0.8 actions) free-running for 200 consecutive
    substeps, fp32 kernels vs fp64 oracle, every substep compared; plus ALL 14 shapes x 3 poses x 4 starts with the grasp-and-lift
    script, asserted PER SHAPE (tests/studies/long_horizon.py; profiles/r04_long_horizon.txt has the per-phase histogram of the first
    divergences with qvel / normal-force traces).
    Random actions: >= 98 % of the envs are within 1e-4 at substep 200 and never left it on the way (measured 0.992 of 256; median 9.5e-8).
    Grasp-and-lift scripts: 146 of 168 (round 4: 99).  Cubes: 10 - 12 of 12.  Round shapes (67-gon cylinders / vases, cones) are where MuJoCo's own
    contact model is discontinuous: the single MPR contact of a finger on a polygonal "round" surface jumps from one facet to the next
    (normals 5.4 degrees apart) and a resting rim has 67 equally deep vertices - an fp32 state error of 1e-7 decides such an event one
    substep earlier or later and the trajectories then differ by 1e-3 - 1e-2.  (What real MuJoCo does in such events is decided by ties of
    its support functions - DESIGN.md section 2, rows 46-62 of the recorded trajectory; the earlier claim that row 22 of that recording was
    such an event is retracted: it was the command recovery.)"""
    from tests.studies import long_horizon as lh
    res = lh.config2_batch(256, 200)
    rel = res["rel"]
    fb = lh.first_bad(rel)
    never = float(np.mean(fb < 0))
    print(lh.summarize("config 2 x 256", res))
    assert (res["status"] == 0).all()
    assert never >= 0.98 and np.median(rel[199]) < 3e-7 and np.percentile(rel[199], 90) < 1e-6        # (measured 0.992 / 9.5e-8 / 2.4e-7)
    shapes = lh.shapes_batches(4, 200)
    within = {}
    for sh, r in shapes.items():
        print(lh.summarize(sh, r))
        assert np.isfinite(r["rel"]).all() and (r["status"] & 2 == 0).all()
        within[sh] = int((r["rel"][199] <= 1e-4).sum())
    print("envs of 12 within 1e-4 at substep 200:", within, "total", sum(within.values()), "of", 12 * len(within))
    short = {sh: (k, LONG_HORIZON_MEASURED[sh]) for sh, k in within.items() if k < LONG_HORIZON_MEASURED[sh] - 1}       # (per-shape floors at measured - 1)
    assert not short, short
    assert sum(within.values()) >= sum(LONG_HORIZON_MEASURED.values()) - 2 and np.median(shapes["CubeS"]["rel"][199]) < 1e-5


# ... and THROUGH ks_step, the product's own stepping path (15 substeps per call, the lanes' pair memory carried from substep to substep and from call
# to call): measured at the end of round 6 after env-step 14 (210 substeps).  Until round 6 this path was not covered - the test above drives
# ks_substep, whose queries start cold - and its penetration queries started warm from the previous portal: 76 of 168 (tests/studies/long_horizon_envstep.py).
ENVSTEP_LONG_HORIZON_MEASURED = {"CubeS": 12, "CubeB": 12, "CylinderS": 12, "CylinderB": 11, "Cube45S": 11, "Cube45B": 12, "Cone1S": 10, "Cone1B": 12,
                                 "Cone2S": 11, "Cone2B": 11, "Vase1S": 12, "Vase1B": 12, "Vase2S": 12, "Vase2B": 12}


def test_long_horizon_parity_through_the_env_step_path_210_substeps():
    """The same 168 grasp-and-lift envs and the config-2 random-action batch, stepped with ks_step (what KinovaGripperVecEnv.step and - same device
    code - the free-running rollout call) against the oracle's env_step: 14 env-steps = 210 substeps, compared after every env-step.  Per shape
    at most one env below the measured count, two in total."""
    from tests.studies import long_horizon_envstep as le
    r = le.config2_batch(256, 14)
    assert (r["status"] == 0).all()
    never = float(np.mean((r["rel"] <= 1e-4).all(0)))
    print(f"config 2 x 256 through ks_step: never beyond 1e-4 in 210 substeps {never:.3f}, median at the end {np.median(r['rel'][13]):.1e}")
    assert never >= 0.98 and np.median(r["rel"][13]) < 3e-7                      # (measured 0.996 / 9.0e-8)
    res = le.shapes_batches(4, 14)
    within = {sh: int((x["rel"][13] <= 1e-4).sum()) for sh, x in res.items()}
    print("through ks_step, envs of 12 within 1e-4 after 210 substeps:", within, "total", sum(within.values()))
    for sh, x in res.items():
        assert np.isfinite(x["rel"]).all() and (x["status"] & 2 == 0).all()
    short = {sh: (k, ENVSTEP_LONG_HORIZON_MEASURED[sh]) for sh, k in within.items() if k < ENVSTEP_LONG_HORIZON_MEASURED[sh] - 1}
    assert not short, short
    assert sum(within.values()) >= sum(ENVSTEP_LONG_HORIZON_MEASURED.values()) - 2


def test_env_step_path_equals_fifteen_substep_calls(cube):
    """ks_step (15 substeps in one launch, pair memory carried along) against 15 ks_substep calls with the controls the env layer derives from the same
    action (cold queries, nothing remembered): since round 6 the penetration query remembers nothing, so the two paths differ only by the distance query's
    remembered simplex - which does not change what it converges to.  256 CubeS envs, closing grasp + lift script, 12 env-steps = 180 substeps of
    approach, contact and lift.  (With round 5's warm-started penetration query the two paths parted at the first flat contact.)"""
    n, T = 256, 12
    q0, hq = scenarios.config2_states(n)
    script = np.array([[0.0, 0.6, 0.5, 0.7]] * 9 + [[0.6, 0.5, 0.5, 0.5]] * (T - 9))
    o = ko.OracleSim(cube, hq[:, 0].copy(), solver_iterations=SOLVER_ITERATIONS)
    o.env_reset(q0[:, 0].copy())
    a_sim = _sim(n, "CubeS", solver_iterations=SOLVER_ITERATIONS, horizon=0)
    b_sim = _sim(n, "CubeS", solver_iterations=SOLVER_ITERATIONS, horizon=0)
    for s_ in (a_sim, b_sim):
        s_.reset(torch.as_tensor(q0), torch.as_tensor(hq))
    worst = []
    for t in range(T):
        a = np.repeat(script[t][:, None], n, 1)
        a_sim.step(torch.as_tensor(a))
        ctrl = ko.env_ctrl(o.view("geom_xpos").reshape(-1, 3)[1], o.view("geom_xmat").reshape(-1, 9)[1], script[t])[2]       # (the hand's rotation is constant)
        ct = torch.as_tensor(np.repeat(ctrl[:, None], n, 1))
        for _ in range(15):
            b_sim.substep(ct)
        qa = a_sim.get_state()["qpos"].double().cpu().numpy()
        qb = b_sim.get_state()["qpos"].double().cpu().numpy()
        rel = np.abs(qa - qb).max(0) / np.maximum(1e-3, np.abs(qb).max(0))
        worst.append(rel)
    rel = np.array(worst)
    ncon = a_sim.get_state()["ncon"].float().mean().item()
    print(f"ks_step vs 15 x ks_substep over {T} env-steps: median {np.median(rel[-1]):.1e}, p95 {np.percentile(rel[-1], 95):.1e}, max {rel.max():.1e}; "
          f"envs within 1e-5 / 1e-4 at the end {np.mean(rel[-1] <= 1e-5):.3f} / {np.mean(rel[-1] <= 1e-4):.3f}; contacts per env at the end {ncon:.2f}")
    assert np.median(rel[-1]) < 1e-6 and np.mean(rel[-1] <= 1e-4) >= 0.97
    a_sim.close(); b_sim.close()


def test_fp64_kernels_track_the_oracle_free_running_for_200_substeps():
    """What remains of the long-horizon gap when rounding is taken away: the fp64 instantiation of the SAME kernels, free running
    (no teacher forcing) for 200 consecutive substeps on all 168 grasp-and-lift envs of the test above - 14 shapes x 3 poses x 4
    starts, first touches, facet jumps, rim ties and all - stays within 1e-9 relative of the oracle in EVERY env at EVERY substep
    (measured: worst 5e-12 at substep 200).  The fp32 figure above (146 of 168) is therefore rounding amplified by the contact
    model's discontinuities, not a difference of algorithm."""
    from tests.studies import long_horizon as lh
    worst = {}
    for sh, r in lh.shapes_batches(4, 200, precision=64).items():
        assert (r["status"] == 0).all()
        worst[sh] = float(r["rel"].max())
    print("fp64 kernels vs oracle, free running 200 substeps, worst relative qpos error per shape:", {k: f"{v:.1e}" for k, v in worst.items()})
    assert max(worst.values()) < 1e-9, worst


def test_time_limit_done_and_auto_reset():
    n = 64
    q0, hq = scenarios.config2_states(n)
    sim = _sim(n, "CubeS", horizon=3, auto_reset=True)
    obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq)).clone()
    a = torch.zeros(4, n)
    for t in range(3):
        obs, rew, done, info = sim.step(a)
        torch.cuda.synchronize()
        assert ((done & 2) != 0).all().item() == (t == 2)
    # after the time limit every env is back at its own initial state and `obs` is the reset observation
    torch.testing.assert_close(obs, obs0, rtol=0, atol=0)
    st = sim.get_state()
    np.testing.assert_allclose(st["qpos"].cpu().numpy(), q0.astype(np.float32), rtol=0, atol=0)
    assert st["qvel"].abs().max().item() == 0
    # the terminal observation of the finished episode is kept in final_obs and differs from the reset one
    assert (sim.final_obs - obs0).abs().max().item() > 1e-4
    sim.close()


def test_full_size_properties():
    """BASELINE size (4096 envs): size-independent properties instead of a per-env oracle run."""
    n = 4096
    q0, hq = scenarios.config2_states(n)
    acts = torch.as_tensor(scenarios.config_actions(64, 4)).repeat(1, 1, n // 64)
    outs = []
    for rep in range(2):
        sim = _sim(n, "CubeS")
        sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
        for t in range(4):
            obs, rew, done, info = sim.step(acts[t])
        torch.cuda.synchronize()
        st = sim.get_state()
        outs.append((obs.clone(), st["qpos"].clone(), st["status"].clone()))
        sim.close()
    # determinism: bit-identical across two runs
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    obs, qpos, status = outs[0]
    assert torch.isfinite(obs).all() and torch.isfinite(qpos).all()
    assert (status & 2).sum().item() == 0
    # envs i and i+64k share the action stream but start elsewhere: results must differ (no aliasing)
    assert (qpos[:, 0] - qpos[:, 64]).abs().max().item() > 1e-6
    # tendon equality keeps distal ~ proximal/2 (XML:171-188)
    prox, dist = qpos[3:9:2], qpos[4:9:2]
    assert (prox - 2 * dist).abs().max().item() < 0.05
    # object quaternion stays normalised
    assert (qpos[12:16].norm(dim=0) - 1).abs().max().item() < 1e-5


def test_naive_demonstrator_lifts_centred_cubes():
    """Behavioural pin: the naive controller lifts a centred CubeS within the 30-step horizon;
    the reference's demonstrations succeed in 21-28 steps."""
    from kinovagrasping_amd.demonstrators import run_naive_episodes
    from kinovagrasping_amd.replay import DeviceEpisodeReplay
    n = 64
    q0 = np.zeros((16, n)); q0[12] = 1; q0[11] = 0.0654
    q0[9] = np.linspace(-0.01, 0.01, n)            # object x across the palm centre
    hq = np.repeat(scenarios.hand_quat_for("normal")[:, None], n, 1)
    sim = _sim(n, "CubeS", horizon=30, auto_reset=False)
    obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
    rep = DeviceEpisodeReplay(n, capacity=128, device=sim.device)
    out = run_naive_episodes(sim, obs0, rep)
    torch.cuda.synchronize()
    rate = out["success"].float().mean().item()
    st = out["steps"][out["success"]].float()
    print(f"naive controller: success {rate:.2f}, steps to lift min {st.min().item():.0f} mean {st.mean().item():.1f} max {st.max().item():.0f}")
    assert rate >= 0.8
    assert 18 <= st.mean().item() <= 30
    assert rep.count == n and (rep.ep_reward[:n].sum(1) >= 50 * out["success"].float().cpu().to(rep.device)).all()
    sim.close()


def test_mixed_shape_batch_equals_per_shape_contexts(assets_dir):
    """BASELINE config 5 plumbing: a mixed-object batch with per-env randomised object mass / friction is bit-identical
    to running each shape alone, and each block agrees with the oracle for its own shape and parameters."""
    from kinovagrasping_amd.multi_shape import MultiShapeSim
    shapes = ["CubeB", "CylinderS", "Cone1B", "Vase2S"]
    per = 32
    n = per * len(shapes)
    rng = np.random.RandomState(5)
    q0 = np.zeros((16, n)); q0[12] = 1
    for k, sh in enumerate(shapes):
        tab = scenarios.start_coord_table(sh)
        q0[9:12, k * per:(k + 1) * per] = tab[rng.randint(0, len(tab), per)].T
    hq = np.repeat(scenarios.hand_quat_for("normal")[:, None], n, 1)
    acts = torch.as_tensor(scenarios.config_actions(n, 2, base_seed=77))
    mass, mu = scenarios.config5_env_params(n)
    ms = MultiShapeSim(n, shapes)
    ms.set_env_params(mass, mu)
    obs_m = ms.reset(torch.as_tensor(q0), torch.as_tensor(hq)).clone()
    for t in range(2):
        om, rm, dm, im = ms.step(acts[t])
    torch.cuda.synchronize()
    qm = ms.get_state()["qpos"]
    for k, sh in enumerate(shapes):
        sl = slice(k * per, (k + 1) * per)
        sim = _sim(per, sh)
        sim.set_env_params(mass[sl], mu[sl])
        o0 = sim.reset(torch.as_tensor(q0[:, sl]), torch.as_tensor(hq[:, sl]))
        assert torch.equal(o0, obs_m[sl])
        for t in range(2):
            o1 = sim.step(acts[t][:, sl])[0]
        torch.cuda.synchronize()
        assert torch.equal(o1, om[sl]) and torch.equal(sim.get_state()["qpos"], qm[:, sl])
        # oracle for this shape, first env of the block
        model = ko.OracleModel(scenarios.model_blob(sh))
        o = ko.OracleSim(model, hq[:, 0], solver_iterations=SOLVER_ITERATIONS)
        o.s.obj_mass, o.s.obj_mu = mass[k * per], mu[k * per]
        o.env_reset(q0[:, k * per])
        for t in range(2):
            o.env_step(acts[t][:, k * per].numpy())
        qo = o.view("qpos")
        rel = np.abs(qm[:, k * per].double().cpu().numpy() - qo).max() / max(1e-3, np.abs(qo).max())
        assert rel < 1e-4, (sh, rel)
        sim.close()
    ms.close()


def test_rollout_kernels_equal_torch_bookkeeping():
    """The kr_* kernels (include/kinova_rollout.h) against the torch implementation of the same rules
    (RolloutEngine.pre/post, DeviceEpisodeReplay) on identical synthetic sim outputs: every piece of engine /
    replay state and a sampled batch must be bit-identical."""
    from types import SimpleNamespace
    from kinovagrasping_amd.replay import DeviceEpisodeReplay
    from kinovagrasping_amd.rollout import RolloutEngine
    dev = torch.device("cuda", 0)
    n, T = 193, 75
    g = torch.Generator(device=dev).manual_seed(5)

    class FakeSim:
        def __init__(self):
            self.n_envs, self.device = n, dev
            self.cfg = SimpleNamespace(auto_reset=1)
            self.obs = torch.zeros(n, 82, device=dev); self.final_obs = torch.zeros(n, 82, device=dev)
            self.reward = torch.zeros(n, device=dev); self.done = torch.zeros(n, dtype=torch.uint8, device=dev)

    W1 = torch.randn(82, 4, device=dev, generator=g) * 0.05
    policy = SimpleNamespace(actor=lambda o: 0.8 * torch.sigmoid(o @ W1))
    engines = []
    for native in (True, False):
        sim = FakeSim()
        rep = DeviceEpisodeReplay(n, capacity=256, horizon=30, device=dev)
        rep.native = native
        eng = RolloutEngine(sim, policy, rep, expl_noise=0.1, generator=torch.Generator(device=dev).manual_seed(11))
        eng.native = native
        engines.append((sim, rep, eng))
    obs0 = torch.randn(n, 82, device=dev, generator=g) * 0.1
    for _, _, eng in engines:
        eng.start(obs0)
    age = torch.zeros(n, dtype=torch.long, device=dev)
    for step in range(T):
        # synthetic sim outputs: fingertips freeze for some envs (check_grasp fires), episodes end at random or at 30
        nobs = torch.randn(n, 82, device=dev, generator=g) * 0.1
        frozen = torch.rand(n, device=dev, generator=g) < 0.3
        nobs[:, 9:17] = torch.where(frozen.unsqueeze(1), engines[0][2].obs[:, 9:17], nobs[:, 9:17])
        fin = torch.randn(n, 82, device=dev, generator=g)
        rew = torch.rand(n, device=dev, generator=g) * 50
        age += 1
        done = (torch.rand(n, device=dev, generator=g) < 0.04) | (age >= 30)
        age = torch.where(done, torch.zeros_like(age), age)
        for sim, rep, eng in engines:
            eng.pre()
            sim.obs.copy_(nobs); sim.final_obs.copy_(fin); sim.reward.copy_(rew); sim.done.copy_(done.to(torch.uint8) * 3)
            eng.post()
        (sa, ra, ea), (sb, rb, eb) = engines
        for name in ("obs", "prev_obs", "has_prev", "t", "ready", "lifting", "action", "action_t", "reward_out", "done_out"):
            assert torch.equal(getattr(ea, name), getattr(eb, name)), (step, name)
        for name in ("cur_state", "cur_next", "cur_action", "cur_reward", "cur_not_done", "cur_len", "_head", "_count"):
            assert torch.equal(getattr(ra, name), getattr(rb, name)), (step, name)
        assert torch.equal(ra.ep_len[:ra.capacity], rb.ep_len[:rb.capacity]), step      # the row behind the ring is torch's trash row
    (sa, ra, ea), (sb, rb, eb) = engines
    assert ra.count > 100 and ea.lifting.any()
    cap = ra.capacity
    for name in ("ep_state", "ep_next", "ep_action", "ep_reward", "ep_not_done"):
        assert torch.equal(getattr(ra, name)[:cap], getattr(rb, name)[:cap]), name
    u = torch.rand(64 * 26, device=dev, generator=g)
    for x, y in zip(ra.sample_batch_nstep(64, uniforms=u), rb.sample_batch_nstep(64, uniforms=u)):
        assert torch.equal(x, y)


@pytest.mark.parametrize("hidden,form", [((256, 256), "lds_free"), ((400, 300), "library_gemm")])
def test_native_learner_update_equals_autograd_update(hidden, form):
    """learner_native against DDPGfD.train_on_batch (autograd + torch.optim.Adam): same losses and the same parameters /
    targets after several updates on masked fixed-shape batches - in both of its forms: the LDS-free MFMA kernels
    (BASELINE widths 256-256) and the library GEMMs + kr_* glue kernels (the reference's 400-300)."""
    from kinovagrasping_amd.ddpgfd import DDPGfD
    from kinovagrasping_amd.learner_native import NativeDDPGfDUpdate
    dev = torch.device("cuda", 0)
    torch.manual_seed(7)
    pa = DDPGfD(82, 4, 0.8, 5, hidden=hidden, device=dev)
    torch.manual_seed(7)
    pb = DDPGfD(82, 4, 0.8, 5, hidden=hidden, device=dev)
    nat = NativeDDPGfDUpdate(pb)
    assert nat.lds_free == (form == "lds_free") and nat.fused_targets
    g = torch.Generator(device=dev).manual_seed(3)
    R, n = 320, 5
    for it in range(12):                      # crosses the soft target update of the 10th call
        st = torch.randn(R, n, 82, device=dev, generator=g) * 0.3
        ns = torch.randn(R, n, 82, device=dev, generator=g) * 0.3
        ac = torch.rand(R, n, 4, device=dev, generator=g) * 0.8
        rw = torch.rand(R, n, device=dev, generator=g) * 5
        w = (torch.rand(R, device=dev, generator=g) < 0.8).float()
        la = pa.train_on_batch(st, ac, ns, rw, w)
        lb = nat.train_on_batch(st, ac, ns, rw, w)
        for x, y in zip(la[1:], lb):
            assert abs(x.item() - y.item()) <= 2e-4 * max(1.0, abs(x.item())), (it, x.item(), y.item())
    for name in ("critic", "actor", "critic_target", "actor_target"):
        a, b = pa._flat_params[name], pb._flat_params[name]
        err = (a - b).abs().max().item()
        print(f"{name}: max |autograd - native| after 12 updates = {err:.2e} (max |param| {a.abs().max().item():.2f})")
        # Adam moves every entry by ~lr per update whatever the gradient's size, so entries whose gradient is rounding
        # noise may drift apart by a fraction of lr * updates; the bound is 10 % of that path length
        lr = 1e-3 if name.startswith("critic") else 1e-4
        assert err <= 0.1 * lr * 12, name
    assert (pa._flat_params["critic_target"] - pa._flat_params["critic"]).abs().max().item() > 0


def test_graphed_trainer_runs_rollout_and_updates():
    """pipeline.GraphedTrainer end to end on 256 envs: the HIP-graph replay of pre / sim / learner / post advances
    episodes, fills the replay, updates both networks and keeps everything finite."""
    from kinovagrasping_amd.ddpgfd import DDPGfD
    from kinovagrasping_amd.pipeline import GraphedTrainer
    from kinovagrasping_amd.replay import DeviceEpisodeReplay
    from kinovagrasping_amd.rollout import RolloutEngine
    n = 256
    q0, hq = scenarios.config2_states(n)
    sim = _sim(n, "CubeS", horizon=30, auto_reset=True)
    obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
    torch.manual_seed(2)
    policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=sim.device)
    w0 = policy._flat_params["actor"].clone(), policy._flat_params["critic"].clone(), policy._flat_params["critic_target"].clone()
    replay = DeviceEpisodeReplay(n, capacity=1024, horizon=30, device=sim.device)
    eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
    eng.start(obs0)
    tr = GraphedTrainer(sim, policy, replay, eng, batch_episodes=16)
    tr.capture()
    ep_done = 0
    for _ in range(70):
        rew, done = tr.step()
        ep_done += int(done.sum())
    tr.flush()
    torch.cuda.synchronize()
    assert tr.updates >= 35 and ep_done >= 2 * n
    assert replay.count >= n                               # every env committed at least one episode
    for w_before, name in zip(w0, ("actor", "critic", "critic_target")):
        w = policy._flat_params[name]
        assert torch.isfinite(w).all() and (w - w_before).abs().max().item() > 0, name
    assert torch.isfinite(tr.native.losses).all()
    assert (sim.get_state()["status"] & 2).sum().item() == 0
    sim.close()


def test_graphed_trainer_at_the_metric_shape_is_reproducible():
    """BASELINE config 3 at ITS shape - 4096 envs, 256-256, 64-episode batches - through pipeline.GraphedTrainer for 70 env-steps,
    twice from the same seeds: finite, no status flag beyond the Newton-cap bit, and the two runs end with the SAME weights and env
    states to the bit (no atomics, no launch-order dependence in the stepping kernel, the ray pool, the window sampler or the
    LDS-free learner beside it) - the single-GPU half of the replica-checksum invariant of SURVEY 8e."""
    from kinovagrasping_amd.ddpgfd import DDPGfD
    from kinovagrasping_amd.pipeline import GraphedTrainer
    from kinovagrasping_amd.replay import DeviceEpisodeReplay
    from kinovagrasping_amd.rollout import RolloutEngine
    n = 4096
    q0, hq = scenarios.config2_states(n)

    def run():
        sim = _sim(n, "CubeS", horizon=30, auto_reset=True)
        obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
        torch.manual_seed(2)
        policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=sim.device)
        replay = DeviceEpisodeReplay(n, capacity=4 * n, horizon=30, device=sim.device)
        eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
        eng.start(obs0)
        tr = GraphedTrainer(sim, policy, replay, eng, batch_episodes=64)
        tr.capture()
        for _ in range(70):
            tr.step()
        tr.flush(finish_update=True)
        torch.cuda.synchronize()
        st = sim.get_state()
        out = {k: policy._flat_params[k].clone() for k in ("actor", "critic", "actor_target", "critic_target")}
        out.update(qpos=st["qpos"].clone(), status=st["status"].clone(), count=torch.tensor(replay.count), updates=torch.tensor(tr.updates),
                   losses=tr.native.losses.clone())
        sim.close()
        return out

    a, b = run(), run()
    assert int(a["updates"]) >= 38 and int(a["count"]) >= n
    assert all(torch.isfinite(a[k]).all() for k in ("actor", "critic", "qpos", "losses"))
    assert int((a["status"] & 7).sum()) == 0
    for k in a:
        assert torch.equal(a[k], b[k]), k
    chk = [float(a[k].double().sum()) for k in ("actor", "critic", "actor_target", "critic_target")]
    print("4096-env trainer, 70 steps: weight checksums", [f"{c:.9e}" for c in chk], "identical in two runs")


def test_randomised_mass_and_friction_match_oracle(cube):
    """BASELINE config 5 extension: per-env object mass U[0.05, 0.15] kg and object-hand friction U[0.5, 1.0]
    (ks_set_env_params) against the oracle with the same overrides: a closing + lifting grasp, teacher-forced one
    substep at a time so that every state along the way is compared (fp64 kernels: algorithm-exact; fp32: product)."""
    rng = np.random.Generator(np.random.PCG64(5))
    n_env = 6
    masses, mus = rng.uniform(0.05, 0.15, n_env), rng.uniform(0.5, 1.0, n_env)
    hq = scenarios.hand_quat_for("normal")
    recs, env_of = [], []
    for e in range(n_env):
        s = ko.OracleSim(cube, hq, solver_iterations=SOLVER_ITERATIONS)
        s.s.obj_mass, s.s.obj_mu = masses[e], mus[e]
        q0 = np.zeros(16); q0[9:12] = [0.01 * (e - 2), 0.0, 0.0654]; q0[12] = 1
        s.env_reset(q0)
        ctrl = np.zeros(9); ctrl[5] = 0.2932; ctrl[6:9] = 0.5
        for i in range(330):
            if i == 250:
                ctrl[4] = 0.5
            before = (s.view("qpos").copy(), s.view("qvel").copy(), s.view("qacc_warmstart").copy())
            s.step(ctrl)
            if i % 3 == 0:
                recs.append((before, ctrl.copy(), (s.view("qpos").copy(), s.view("qvel").copy()), s.s.ncon))
                env_of.append(e)
    n = len(recs)
    env_of = np.array(env_of)
    assert max(r[3] for r in recs) >= 4                      # the grasps do make finger contacts
    for precision, tol_med, tol_p99 in ((64, 1e-10, 1e-8), (32, 5e-7, 2e-4)):
        sim = _sim(n, "CubeS", precision=precision, solver_iterations=SOLVER_ITERATIONS)
        q0 = np.stack([r[0][0] for r in recs], 1)
        sim.reset(torch.as_tensor(q0), torch.as_tensor(np.repeat(hq[:, None], n, 1)))
        sim.set_env_params(masses[env_of], mus[env_of])
        sim.set_state(torch.as_tensor(q0), torch.as_tensor(np.stack([r[0][1] for r in recs], 1)), torch.as_tensor(np.stack([r[0][2] for r in recs], 1)))
        sim.substep(torch.as_tensor(np.stack([r[1] for r in recs], 1)))
        st = sim.get_state()
        torch.cuda.synchronize()
        eq = np.abs(st["qpos"].double().cpu().numpy() - np.stack([r[2][0] for r in recs], 1)).max(0)
        print(f"randomised mass/mu, precision {precision}: one-step |dqpos| median {np.median(eq):.2e} p99 {np.percentile(eq, 99):.2e} max {eq.max():.2e}")
        assert np.median(eq) <= tol_med and np.percentile(eq, 99) <= tol_p99
        # the overrides matter: the same states stepped with nominal parameters differ from the oracle where there is contact
        if precision == 64:
            sim2 = _sim(n, "CubeS", precision=64, solver_iterations=SOLVER_ITERATIONS)
            sim2.reset(torch.as_tensor(q0), torch.as_tensor(np.repeat(hq[:, None], n, 1)))
            sim2.set_state(torch.as_tensor(q0), torch.as_tensor(np.stack([r[0][1] for r in recs], 1)), torch.as_tensor(np.stack([r[0][2] for r in recs], 1)))
            sim2.substep(torch.as_tensor(np.stack([r[1] for r in recs], 1)))
            e2 = np.abs(sim2.get_state()["qpos"].double().cpu().numpy() - np.stack([r[2][0] for r in recs], 1)).max(0)
            assert e2.max() > 100 * max(eq.max(), 1e-12)
            sim2.close()
        sim.close()


def test_scripted_demonstrators_grasp_across_the_start_table():
    """Behavioural pin for the expert-data generators: naive, position-dependent and combined
    controllers on 512 CubeS starts spread over the no-noise table; the reference reports most demonstrations
    succeeding within 21-28 steps."""
    from kinovagrasping_amd.demonstrators import run_controller_episodes
    from kinovagrasping_amd.replay import DeviceEpisodeReplay
    n = 512
    tab = scenarios.start_coord_table("CubeS")
    idx = np.linspace(0, len(tab) - 1, n).astype(int)
    q0 = np.zeros((16, n)); q0[12] = 1; q0[9:12] = tab[idx].T
    hq = np.repeat(scenarios.hand_quat_for("normal")[:, None], n, 1)
    rates = {}
    for mode in ("naive", "position-dependent", "combined"):
        sim = _sim(n, "CubeS", horizon=30, auto_reset=False)
        obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
        rep = DeviceEpisodeReplay(n, capacity=n, device=sim.device)
        out = run_controller_episodes(sim, obs0, rep, mode=mode)
        torch.cuda.synchronize()
        rates[mode] = out["success"].float().mean().item()
        st = out["steps"][out["success"]].float()
        print(f"{mode:20s} success {rates[mode]:.2f}  steps to lift mean {st.mean().item():.1f}")
        assert rep.count == n
        sim.close()
    # 0.64 - 0.66 since round 4 (0.91 before): with the explicit pairs at MuJoCo's margin 0 the near-palm starts fail as they do in the
    # reference's own recorded heat map, whose cells are 60 % successes (645 of 1083: tests/test_mujoco_recorded.py)
    assert min(rates.values()) >= 0.55 and rates["combined"] >= 0.6


def test_reference_trained_policy_grasps_in_this_simulator():
    """End-to-end behavioural cross-check: a 400-300 actor the reference authors trained on real
    MuJoCo (policies/rl_exp_pretrain_no_grasp_2_7_2021/pre_DDPGfD_kinovaGrip_02_05_21_2324_actor; weights committed as
    tests/golden/ref_policy_cubes_actor.npz) evaluated with evaluate.eval_policy on 512 CubeS starts of the reference's
    no-noise table.  tests/studies/eval_reference_policies.py ran all thirteen 82-d checkpoints of the reference through the CPU
    oracle: they are pre-training snapshots of very uneven quality (0 - 0.88 success); the two best reach 0.88 here."""
    from pathlib import Path
    from kinovagrasping_amd.ddpgfd import DDPGfD
    from kinovagrasping_amd.evaluate import eval_policy
    n = 512
    tab = scenarios.start_coord_table("CubeS")
    idx = np.linspace(0, len(tab) - 1, n).astype(int)
    q0 = np.zeros((16, n)); q0[12] = 1; q0[9:12] = tab[idx].T
    hq = np.repeat(scenarios.hand_quat_for("normal")[:, None], n, 1)
    sim = _sim(n, "CubeS", horizon=30, auto_reset=False)
    obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
    pol = DDPGfD(82, 4, 0.8, 5, hidden=(400, 300), device=sim.device)
    w = np.load(Path(__file__).resolve().parent / "golden" / "ref_policy_cubes_actor.npz")
    pol.actor.load_state_dict({k.replace("_", ".", 1): torch.as_tensor(w[k]) for k in w.files})
    out = eval_policy(sim, pol, obs0)
    rate = out["num_success"] / n
    print(f"reference-trained actor in this sim: success {rate:.2f}, avg reward {out['avg_reward']:.1f}, "
          f"mean steps of successes {out['steps'][out['success']].float().mean().item():.1f}")
    assert len(out["success_coords"]["x"]) + len(out["fail_coords"]["x"]) == n
    assert abs(out["avg_rewards"]["lift_reward"] - 50 * rate) < 1e-3 and out["avg_rewards"]["finger_reward"] == 0
    assert rate >= 0.7
    sim.close()


def test_all_fourteen_shapes_track_the_oracle(assets_dir):
    """Every README object (14 shapes) x 8 starts: 6 env-steps (90 substeps) of a closing grasp on the GPU against
    the fp64 oracle run on the same starts and actions, no status flags.
      * fp64 kernel (the same source as the fp32 one): at least 6 of the 8 envs agree to 1e-9 relative qpos error -
        this is the check of the kernel LOGIC (two envs may leave the oracle's trajectory: the GJK warm start ends
        on a different, equally valid simplex once in a while and a contact-rich grasp amplifies that to ~5e-3);
      * fp32 kernel: rounding only.  Per shape the median relative error is <= 5e-5 and at most two of the eight
        envs exceed 2e-4 (grasps whose contact set flipped: SURVEY hard part 2)."""
    per = 8
    worst = {}
    bad = []
    for sh in scenarios.SHAPES:
        tab = scenarios.start_coord_table(sh)
        idx = np.linspace(0, len(tab) - 1, per).astype(int)
        q0 = np.zeros((16, per)); q0[12] = 1; q0[9:12] = tab[idx].T
        hq = np.repeat(scenarios.hand_quat_for("normal")[:, None], per, 1)
        act = np.repeat(np.array([0.0, 0.6, 0.5, 0.7])[:, None], per, 1)
        qg = {}
        for prec in (32, 64):
            sim = _sim(per, sh, precision=prec)
            sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
            for t in range(6):
                sim.step(torch.as_tensor(act))
            torch.cuda.synchronize()
            st = sim.get_state()
            qg[prec] = st["qpos"].double().cpu().numpy()
            assert (st["status"].cpu().numpy() == 0).all(), (sh, prec)
            sim.close()
        model = ko.OracleModel(scenarios.model_blob(sh))
        rel = {32: [], 64: []}
        for i in range(per):
            o = ko.OracleSim(model, hq[:, i], solver_iterations=SOLVER_ITERATIONS)
            o.env_reset(q0[:, i])
            for t in range(6):
                o.env_step(act[:, i])
            qo = o.view("qpos")
            for prec in (32, 64):
                rel[prec].append(np.abs(qg[prec][:, i] - qo).max() / max(1e-3, np.abs(qo).max()))
        r32, r64 = np.array(rel[32]), np.array(rel[64])
        worst[sh] = (float(np.median(r32)), float(r32.max()), float(np.sort(r64)[-2]))
        if np.median(r32) > 5e-5 or (r32 <= 2e-4).sum() < per - 2 or (r64 <= 1e-9).sum() < per - 2:
            bad.append((sh, r32, r64))
    assert not bad, bad


def test_vec_env_keeps_the_reference_interface():
    """KinovaGripperVecEnv: reset(shape_keys, hand_orientation, ...) / step(action) with the reference's return layout
    (kinova_gripper_env.py:1310, 1495, 685), random orientation classes, partial resets."""
    from kinovagrasping_amd.vec_env import KinovaGripperVecEnv
    n = 96
    env = KinovaGripperVecEnv(n, "CylinderB", seed=3, auto_reset=False, hand_offsets="pose")
    obs = env.reset(shape_keys=["CylinderB"], hand_orientation="random", with_grasp=False, mode="train", with_noise=False)
    assert tuple(obs.shape) == (n, 82) and torch.isfinite(obs).all()
    assert set(env.get_orientation()) == {"normal", "rotated", "top"}
    assert env.action_space.shape == (4,) and env._max_episode_steps == 30
    a = torch.zeros(n, 4); a[:, 1:] = 0.5
    for _ in range(3):
        obs, reward, done, info = env.step(a)
    assert tuple(reward.shape) == (n,) and done.dtype == torch.bool
    assert set(info) >= {"finger_reward", "grasp_reward", "lift_reward"} and (info["finger_reward"] == 0).all()
    # object start coordinates come from the table of each env's orientation class
    for e in (0, n // 2, n - 1):
        tab = scenarios.start_coord_table("CylinderB", env.get_orientation()[e])
        assert (np.abs(tab - env.get_obj_coords()[e]).sum(1) < 1e-12).any()
    before = obs.clone()
    ids = [1, 5, 17]
    env.reset(hand_orientation="normal", env_ids=ids, with_noise=False)
    torch.cuda.synchronize()
    changed = (env.sim.obs != before).any(1).cpu().numpy()
    assert changed[ids].all() and changed.sum() == len(ids)
    with pytest.raises(NotImplementedError):
        env.set_with_grasp_reward(True)
    env.close()


def test_vec_env_reset_test_hooks_of_the_reference():
    """reset()'s test hooks as the reference has them (ENV:1310-1363, 1029-1048, 1171-1178, 349-353): obj_params [shape, size], start_pos rows of 3 / 2 /
    9 values, obj_coord_region (incl. the reference's index slip: the row index is drawn among the region's rows and used on the whole file), qpos."""
    from kinovagrasping_amd.vec_env import KinovaGripperVecEnv
    n = 12
    env = KinovaGripperVecEnv(n, ["CubeS", "CylinderB", "Vase2S"], seed=21, auto_reset=False)
    env.reset(obj_params=["Cylinder", "B"])                          # (all defaults: the reference's with_noise tables, zero hand offsets)
    assert all("with_noise/train_coords" in f for f in env.get_coords_filename())
    assert env.get_random_shape() == ["CylinderB"] * n
    with pytest.raises(ValueError):
        env.reset(obj_params=["Cone1", "S"])
    size_b = float(mc_read("CylinderB")["obj_size_obs"][2]) / 2
    sp = [[0.01, 0.02, 0.07]] * 4 + [[-0.02, 0.03]] * 4 + [[0.0, 0.0, 0.01, 0.3, 0.2, 0.1, 0.02, 0.01, 0.066]] * 4
    env.reset(obj_params=["Cylinder", "B"], start_pos=sp)
    q = env.sim.get_state()["qpos"].double().cpu().numpy()
    assert np.allclose(q[9:12, 0], [0.01, 0.02, 0.07], atol=1e-6) and np.allclose(q[9:12, 5], [-0.02, 0.03, size_b], atol=1e-6)
    assert np.allclose(q[[0, 1, 2, 3, 5, 7], 9], [0.0, 0.0, 0.01, 0.3, 0.2, 0.1], atol=1e-6) and np.allclose(q[9:12, 9], [0.02, 0.01, 0.066], atol=1e-6)
    assert np.allclose(env.get_obj_coords()[5], [-0.02, 0.03, size_b])
    # regions
    tab = scenarios.start_coord_table("CubeS", "normal")
    for region, (lo, hi) in {"left": (-.09, -.03), "center": (-.03, .03), "target": (-.01, .01), "right": (.03, .09)}.items():
        env.reset(obj_params=["Cube", "S"], obj_coord_region=region, with_noise=False)
        count = int(((tab[:, 0] >= lo) & (tab[:, 0] <= hi)).sum())
        idx = np.asarray(env.get_orientation_idx())
        assert (idx >= 0).all() and (idx < count).all()                          # drawn among the region's rows ...
        assert all(np.array_equal(env.get_obj_coords()[e], tab[idx[e]]) for e in range(n))      # ... and used on the whole file (sic)
    env.reset(obj_params=["Cube", "S"], obj_coord_region="origin", with_noise=False)
    assert np.allclose(env.get_obj_coords(), np.array([0.0, 0.0, tab[0][2]])[None])
    # qpos: the given joint vector as it is
    qq = np.zeros((n, 16)); qq[:, 12] = 1; qq[:, 9:12] = [0.0, 0.01, 0.08]; qq[:, 3] = 0.25; qq[:, 4] = 0.125
    obs = env.reset(obj_params=["Cube", "S"], qpos=qq, start_pos=[[0.0, 0.01, 0.08]] * n)
    q = env.sim.get_state()["qpos"].double().cpu().numpy()
    assert np.allclose(q, qq.T, atol=1e-6) and torch.isfinite(obs).all()
    env.close()


def mc_read(shape):
    from kinovagrasping_amd import model_compiler
    return model_compiler.read_blob(scenarios.model_blob(shape))


def test_vec_env_reference_with_noise_tables_mode():
    """reset() with the DEFAULTS (with_noise=True, hand_offsets="fresh-env"): the reference's DEFAULT start states as they are (ENV:1310, 1019-1021, 1254-1255) - object position AND hand
    Euler triple of a random row of the shape's with_noise file, truncated to 5 characters; every env's reset observation equals the oracle's
    at that pose.  (With hand_offsets="fresh-env" - what the reference's training driver ends up with - the 'normal' class of these tables puts
    the hand at the IDENTITY orientation + noise: SURVEY N5.)"""
    from kinovagrasping_amd.vec_env import KinovaGripperVecEnv
    from kinovagrasping_amd.model_compiler import euler_to_quat, truncated_euler
    n = 36
    env = KinovaGripperVecEnv(n, "Cube45B", seed=4, auto_reset=False)
    assert env.hand_offsets == "fresh-env"
    obs = env.reset(["Cube45B"], "random").double().cpu().numpy().copy()          # positional, no keywords: as the reference's drivers call it
    assert set(env.get_orientation()) == {"normal", "rotated", "top"} and all("with_noise/train_coords" in f for f in env.get_coords_filename())
    model = ko.OracleModel(scenarios.model_blob("Cube45B"))
    for e in range(n):
        tab = scenarios.noisy_start_table("Cube45B", env.get_orientation()[e])
        row = tab[env.get_orientation_idx()[e]]
        assert np.array_equal(row[:3], env.get_obj_coords()[e]) and np.array_equal(truncated_euler(row[3:6]), env.hand_euler[e])
        assert np.allclose(env.hand_quat[:, e], euler_to_quat(env.hand_euler[e]))
    worst = 0.0
    for e in range(0, n, 4):
        o = ko.OracleSim(model, env.hand_quat[:, e].copy(), solver_iterations=SOLVER_ITERATIONS)
        q0 = np.zeros(16); q0[9:12] = env.get_obj_coords()[e]; q0[12] = 1
        ref = o.env_reset(q0)
        tol = 2e-4 * np.maximum(1.0, np.abs(ref)) * np.where(np.isin(np.arange(82), [48, 49] + list(range(75, 82))), 50, 1)
        assert (np.abs(obs[e] - ref) <= tol + 2e-5).all(), (e, np.abs(obs[e] - ref).max())
        worst = max(worst, np.abs(obs[e] - ref).max())
    # the class swap of N5, seen through the env: the 'normal' rows hold near-identity hand orientations
    normal = [e for e in range(n) if env.get_orientation()[e] == "normal"]
    assert normal and all(np.abs(env.hand_euler[e]).max() < 0.5 for e in normal)
    print(f"with_noise='tables': worst reset-observation error vs oracle {worst:.1e}")
    env.close()
    with pytest.raises(ValueError):
        KinovaGripperVecEnv(4, "CubeS", auto_reset=False).reset(with_noise="yes")


def test_vec_env_with_noise_resets_to_noisy_poses_that_match_the_oracle():
    """reset(with_noise="zero-mean"): zero-mean N(0, 0.087) Euler noise through the 5-character truncation (scenarios.hand_euler_for,
    SURVEY N5 extension).  Every env's reset observation and its state after two env-steps equal the oracle run on the SAME noisy
    hand quaternion; the noise is there (quaternions differ from the class constant) and seeded."""
    from kinovagrasping_amd.vec_env import KinovaGripperVecEnv
    from kinovagrasping_amd.model_compiler import euler_to_quat
    n = 48
    env = KinovaGripperVecEnv(n, "CubeS", seed=9, auto_reset=False)
    obs = env.reset(hand_orientation="normal", with_noise="zero-mean").double().cpu().numpy().copy()
    base = scenarios.hand_quat_for("normal")
    ang = 2 * np.arccos(np.clip(np.abs(env.hand_quat.T @ base), 0, 1))
    assert ang.min() > 1e-3 and 0.05 < ang.mean() < 0.3                      # a few degrees of tilt in every env
    assert np.allclose(env.hand_quat, np.stack([euler_to_quat(e) for e in env.hand_euler], 1))
    env2 = KinovaGripperVecEnv(n, "CubeS", seed=9, auto_reset=False)
    env2.reset(hand_orientation="normal", with_noise="zero-mean")
    assert np.array_equal(env.hand_euler, env2.hand_euler)
    env2.close()
    model = ko.OracleModel(scenarios.model_blob("CubeS"))
    a = np.array([0.1, 0.5, 0.4, 0.6])
    act = torch.as_tensor(np.repeat(a[None], n, 0))
    for _ in range(2):
        env.step(act)
    qg = env.sim.get_state()["qpos"].double().cpu().numpy()
    worst_o, rel_q = 0.0, []
    for e in range(0, n, 3):
        o = ko.OracleSim(model, env.hand_quat[:, e].copy(), solver_iterations=SOLVER_ITERATIONS)
        q0 = np.zeros(16); q0[9:12] = env.get_obj_coords()[e]; q0[12] = 1
        ref = o.env_reset(q0)
        tol = 2e-4 * np.maximum(1.0, np.abs(ref)) * np.where(np.isin(np.arange(82), [48, 49] + list(range(75, 82))), 50, 1)
        assert (np.abs(obs[e] - ref) <= tol + 2e-5).all(), (e, np.abs(obs[e] - ref).max())
        worst_o = max(worst_o, np.abs(obs[e] - ref).max())
        for _ in range(2):
            o.env_step(a)
        qo = o.view("qpos")
        rel_q.append(np.abs(qg[:, e] - qo).max() / max(1e-3, np.abs(qo).max()))
    rel_q = np.sort(rel_q)
    print(f"noisy poses: worst reset-obs error {worst_o:.2e}, relative qpos error after 2 env-steps: median {np.median(rel_q):.2e}, second worst {rel_q[-2]:.2e}, worst {rel_q[-1]:.2e}")
    # fp32 against the fp64 oracle over 30 substeps of contact: all envs but at most one follow to 1e-4; the one is a discrete event (MPR ends
    # on another facet of the fingertip hull: 1.4e-3 on env 3 with this seed), bounded like the long-horizon study's tail
    assert np.median(rel_q) < 1e-5 and rel_q[-2] < 1e-4 and rel_q[-1] < 5e-3
    env.close()


def test_fused_mlp_forward_matches_torch():
    """kr_mlp3_forward (fp32 MFMA, one launch) against the torch modules it replaces: the actor (82 -> h1 -> h2 -> 4,
    0.8 * sigmoid) and the critic on cat([state, action]) (86 -> h1 -> h2 -> 1), at the BASELINE widths (256-256), the
    reference's (400-300) and a small one, for batch sizes that are not multiples of the 16-row tile, with the inputs
    read through row-strided views.  Same fp32 arithmetic, different summation order: 2e-5 absolute."""
    from kinovagrasping_amd import mlp
    from kinovagrasping_amd.ddpgfd import Actor, Critic
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(7)
    for hidden in ((256, 256), (400, 300), (64, 64)):
        torch.manual_seed(11)
        actor, critic = Actor(82, 4, 0.8, hidden).to(dev), Critic(82, 4, hidden).to(dev)
        for net in (actor, critic):             # biases away from zero, weights large enough to exercise the ReLUs
            for p in net.parameters():
                p.data.add_(0.05 * torch.randn(p.shape, generator=g).to(dev))
        for n in (1, 37, 1000, 4096):
            wide = torch.randn(n, 100, generator=g).to(dev)
            s, a = wide[:, :82], (0.8 * torch.rand(n, 4, generator=g)).to(dev)
            with torch.no_grad():
                ref_a, ref_q = actor(s), critic(s, a)
            h1, h2 = torch.zeros(n, hidden[0], device=dev), torch.zeros(n, hidden[1], device=dev)
            out_a = mlp.mlp3_forward(mlp.layers_of(actor), s, act=mlp.ACT_SIGMOID, scale=0.8, h1_out=h1, h2_out=h2)
            out_q = mlp.mlp3_forward(mlp.layers_of(critic), s, a, act=mlp.ACT_NONE)
            with torch.no_grad():                       # the hidden activations a backward pass would read
                r1 = torch.relu(actor.l1(s)); r2 = torch.relu(actor.l2(r1))
            assert (h1 - r1).abs().max().item() < 2e-5 and (h2 - r2).abs().max().item() < 2e-5, (hidden, n)
            if mlp.supported(mlp.layers_of(actor), 82, shadow=True):        # the LDS-free variants: same results
                import os
                kept = {}
                for split in ("0", "2", "4", None):        # one wave per 16 rows / tiles split over 2 / 4 waves / the default choice
                    if split is None:
                        os.environ.pop("KS_MLP_SPLIT", None)
                    else:
                        os.environ["KS_MLP_SPLIT"] = split
                    g1, g2 = torch.zeros_like(h1), torch.zeros_like(h2)
                    sh_a = mlp.mlp3_forward(mlp.layers_of(actor), s, act=mlp.ACT_SIGMOID, scale=0.8, h1_out=g1, h2_out=g2, shadow=True)
                    sh_q = mlp.mlp3_forward(mlp.layers_of(critic), s, a, act=mlp.ACT_NONE, shadow=True)        # (exchange through scratch)
                    assert (sh_a - ref_a).abs().max().item() < 2e-5 and (sh_q - ref_q).abs().max().item() < 2e-5 * max(1.0, ref_q.abs().max().item()), (hidden, n, split)
                    assert (g1 - r1).abs().max().item() < 2e-5 and (g2 - r2).abs().max().item() < 2e-5, (hidden, n, split)
                    kept[split] = (g1, g2)
                # layers 1 and 2 are the same fma chains in every LDS-free variant: bitwise equal activations
                assert all(torch.equal(kept["0"][0], kept[k][0]) and torch.equal(kept["0"][1], kept[k][1]) for k in ("2", "4"))
            else:
                assert hidden == (400, 300)
            assert out_a.shape == ref_a.shape and out_q.shape == ref_q.shape
            assert (out_a - ref_a).abs().max().item() < 2e-5, (hidden, n, (out_a - ref_a).abs().max().item())
            assert (out_q - ref_q).abs().max().item() < 2e-5 * max(1.0, ref_q.abs().max().item()), (hidden, n, (out_q - ref_q).abs().max().item())
    # unsupported widths are refused, not silently mis-computed
    odd = Actor(82, 4, 0.8, (200, 100)).to(dev)
    assert not mlp.supported(mlp.layers_of(odd), 82)
    with pytest.raises(RuntimeError):
        mlp.mlp3_forward(mlp.layers_of(odd), torch.zeros(4, 82, device=dev), act=mlp.ACT_SIGMOID, scale=0.8)


def test_fused_actor_select_equals_separate_kernels():
    """kr_actor_select (actor forward + noise + selection rule in one launch) against kr_mlp3_forward followed by
    kr_select_action on the same noise tensor: bit-identical actions and latches.  With its in-kernel generator: the
    same seed and counter give the same draws, the counter advances by one per launch, and the draws are N(0,1)
    (mean, variance, 4th moment over 2e5 samples; per-column independence through the correlation matrix)."""
    import ctypes
    from kinovagrasping_amd import mlp, sim as ks
    from kinovagrasping_amd.ddpgfd import Actor
    dev = torch.device("cuda", 0)
    lib, P = ks.load_library(), ks._ptr
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    torch.manual_seed(3)
    n = 5000
    actor = Actor(82, 4, 0.8, (256, 256)).to(dev)
    (w1, b1), (w2, b2), (w3, b3) = mlp.layers_of(actor)
    obs, prev = torch.randn(n, 82, device=dev), torch.randn(n, 82, device=dev)
    prev[: n // 2, 9:16:3] = obs[: n // 2, 9:16:3]                # fingertips at rest: check_grasp fires
    has_prev = torch.rand(n, device=dev) < 0.8
    t = torch.randint(0, 30, (n,), device=dev)
    noise = torch.randn(n, 4, device=dev)

    def fresh():
        return (torch.rand(n, device=dev) < 0.1, torch.zeros(n, 4, device=dev), torch.zeros(4, n, device=dev), torch.zeros(n, dtype=torch.bool, device=dev))

    r1, a1, at1, l1 = fresh()
    r2 = r1.clone(); _, a2, at2, l2 = fresh()
    pi = mlp.mlp3_forward(mlp.layers_of(actor), obs, act=mlp.ACT_SIGMOID, scale=0.8)
    assert lib.kr_select_action(n, P(obs), P(prev), P(has_prev), P(t), P(r1), P(pi), P(noise), 0.08, 0.8, 6, P(a1), P(at1), P(l1), st) == 0
    pi2 = torch.zeros(n, 4, device=dev)
    assert lib.kr_actor_select(n, 256, 256, P(obs), P(prev), P(has_prev), P(t), P(r2), P(w1), P(b1), P(w2), P(b2), P(w3), P(b3), P(noise), 0, None,
                               0.08, 0.8, 6, P(pi2), P(a2), P(at2), P(l2), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(pi, pi2) and torch.equal(a1, a2) and torch.equal(at1, at2) and torch.equal(l1, l2) and torch.equal(r1, r2)
    assert l1.any() and not l1.all()
    # in-kernel noise: zero the actor so that action = clip(0.4 + sigma * z) and recover z (sigma small enough not to clip)
    for p in actor.parameters():
        p.data.zero_()
    zs = []
    rng = torch.zeros(2, dtype=torch.long, device=dev)
    nobody = torch.zeros(n, dtype=torch.bool, device=dev)
    for rep in range(40):
        r, a, at, l = nobody.clone(), torch.zeros(n, 4, device=dev), torch.zeros(4, n, device=dev), nobody.clone()
        assert lib.kr_actor_select(n, 256, 256, P(obs), P(prev), P(nobody), P(t), P(r), P(w1), P(b1), P(w2), P(b2), P(w3), P(b3), None, 1234, P(rng),
                                   0.01, 0.8, 6, None, P(a), P(at), P(l), st) == 0
        zs.append((a - 0.4) / 0.01)
        assert int(rng[0].item()) == rep + 1 and int(rng[1].item()) == 0
    z = torch.cat(zs, 0).double()
    assert abs(z.mean().item()) < 0.01 and abs(z.var().item() - 1.0) < 0.02 and abs((z ** 4).mean().item() - 3.0) < 0.1
    c = torch.corrcoef(z.t())
    assert (c - torch.eye(4, device=dev, dtype=torch.double)).abs().max().item() < 0.01
    assert not torch.equal(zs[0], zs[1])                       # a new draw every launch
    rng2 = torch.zeros(2, dtype=torch.long, device=dev)
    a_again = torch.zeros(n, 4, device=dev)
    assert lib.kr_actor_select(n, 256, 256, P(obs), P(prev), P(nobody), P(t), P(nobody.clone()), P(w1), P(b1), P(w2), P(b2), P(w3), P(b3), None, 1234,
                               P(rng2), 0.01, 0.8, 6, None, P(a_again), P(torch.zeros(4, n, device=dev)), P(nobody.clone()), st) == 0
    assert torch.equal((a_again - 0.4) / 0.01, zs[0])          # same (seed, counter, env) -> same draw


def test_curriculum_stage_runs_and_hands_over_to_the_next(tmp_path):
    """curriculum.run_stage: experiment 1 (sizes: CubeS + CubeB, normal orientation) trained from scratch for one round,
    saved in the reference's directory layout; experiment 4 (sizes x shapes x orientations) then starts from that
    stage's policy and agent replay (rl_experiment, main_DDPGfD.py:776-800).  Plumbing check at toy sizes."""
    from kinovagrasping_amd import curriculum
    from kinovagrasping_amd.ddpgfd import DDPGfD
    dev = torch.device("cuda", 0)
    torch.manual_seed(2)
    policy = DDPGfD(82, 4, 0.8, 5, batch_size=8, hidden=(64, 64), device=dev)
    p1 = curriculum.experiment_plan(1, root=tmp_path)
    # expert demonstrations for the stage's shapes, in the reference's directory layout (expert_replay_data/<grasp>/combined/
    # <shape>/<orientation>/replay_buffer, main_DDPGfD.py:1183-1187): run_stage mixes them in at expert_prob (DDPGfD.py:232-254)
    from kinovagrasping_amd.demonstrators import run_controller_episodes
    from kinovagrasping_amd.replay import DeviceEpisodeReplay
    for shape in ("CubeS", "CubeB"):
        tab = scenarios.start_coord_table(shape)
        q0 = np.zeros((16, 32)); q0[12] = 1; q0[9:12] = tab[np.linspace(0, len(tab) - 1, 32).astype(int)].T
        hq = np.repeat(scenarios.hand_quat_for("normal")[:, None], 32, 1)
        sim = _sim(32, shape, horizon=30, auto_reset=False)
        rep = DeviceEpisodeReplay(32, capacity=64, horizon=30, device=dev)
        res = run_controller_episodes(sim, sim.reset(torch.as_tensor(q0), torch.as_tensor(hq)), replay=rep, mode="combined")
        assert res["success"].float().mean().item() > 0.5
        rep.save(p1["dirs"]["expert_replay_dir"] / shape / "normal" / "replay_buffer")
        sim.close()
    r1 = curriculum.run_stage(p1, policy, n_envs=64, rounds=1, updates_per_round=2, load_previous=False)
    assert r1["shapes"] == ["CubeS", "CubeB"] and r1["skipped_shapes"] == [] and r1["updates"] == 2 and r1["num_total"] == 64
    assert r1["expert_episodes"] >= 40                      # both shapes' demonstrations were found and loaded
    assert curriculum.policy_basename(p1["dirs"]["policy_dir"]).startswith("DDPGfD_kinovaGrip_")
    assert (p1["dirs"]["output_dir"] / "experiment_info.txt").read_text().startswith("NO grasp Experiment 1: sizes, Stage 1")
    w_saved = policy.actor.l1.weight.detach().clone()
    fresh = DDPGfD(82, 4, 0.8, 5, batch_size=8, hidden=(64, 64), device=dev)
    assert not torch.equal(fresh.actor.l1.weight, w_saved)
    p4 = curriculum.experiment_plan(4, root=tmp_path)
    assert p4["dirs"]["prev_policy_dir"] == p1["dirs"]["policy_dir"]
    r4 = curriculum.run_stage(p4, fresh, n_envs=112, rounds=1, updates_per_round=1, save=False)
    # all 14 size x shape keys of the stage run, the multi-geom Bottle / Bowl / TBottle objects included (one context of the
    # multi-geom library holds all of them), in all three orientation classes
    assert r4["shapes"] == ["CubeS", "CylinderS", "Cube45S", "Vase2S", "BottleS", "BowlS", "TBottleS", "CubeB", "CylinderB", "Cube45B", "Vase2B", "BottleB", "BowlB",
                            "TBottleB"]
    assert r4["skipped_shapes"] == [] and r4["num_total"] == 112
    assert set(r4["orientation_counts"]) == {"normal", "rotated", "top"} and r4["updates"] == 1
    # a test-mode stage (TEST_SHAPES: Vase1, RBowl) on the multi-geom library: RBowl never takes the 'normal' class
    pt = curriculum.experiment_plan(5, exp_mode="test", root=tmp_path)
    assert pt["requested_shapes"] == ["Vase1M", "RBowlM"] and pt["requested_orientation"] == "random"
    rt = curriculum.run_stage(pt, fresh, n_envs=32, rounds=1, updates_per_round=1, save=False, load_previous=False)
    assert rt["shapes"] == ["Vase1M", "RBowlM"] and rt["skipped_shapes"] == [] and rt["num_total"] == 32


@pytest.mark.parametrize("split", ["0", "2", "4", None])
def test_lds_free_backward_matches_autograd(split, monkeypatch):
    """kr_mlp3_backward_shadow / _split + kr_weight_grad_shadow against torch autograd on the same networks: the critic's
    gradients for a given dLoss/dQ (all weight / bias gradients and dQ/da), and the actor's for a given dLoss/da
    through the 0.8 * sigmoid output.  fp32, different summation order: 1e-4 relative to the largest entry.
    split: one wave per 16 rows ("0"), the tiles split over 2 / 4 waves of a workgroup, or mlp.py's own choice (None)."""
    from kinovagrasping_amd import mlp
    if split is None:
        monkeypatch.delenv("KS_MLP_SPLIT", raising=False)
    else:
        monkeypatch.setenv("KS_MLP_SPLIT", split)
    from kinovagrasping_amd.ddpgfd import Actor, Critic
    dev = torch.device("cuda", 0)
    for hidden, n in (((256, 256), 1600), ((256, 256), 8000), ((64, 64), 333)):
        torch.manual_seed(5)
        actor, critic = Actor(82, 4, 0.8, hidden).to(dev), Critic(82, 4, hidden).to(dev)
        s = torch.randn(n, 82, device=dev)
        a = (0.8 * torch.rand(n, 4, device=dev)).requires_grad_(True)
        dq = torch.randn(n, 1, device=dev) / n
        q = critic(s, a)
        (q * dq).sum().backward()
        lc = mlp.layers_of(critic)
        h1, h2 = torch.empty(n, hidden[0], device=dev), torch.empty(n, hidden[1], device=dev)
        mlp.mlp3_forward(lc, s, a.detach(), h1_out=h1, h2_out=h2, shadow=True)
        dz2, dz1, da = mlp.mlp3_backward(lc, dq, h1, h2, want_dz=True, dx_cols=(82, 4))
        close = lambda x, y: (x - y).abs().max().item() <= 1e-4 * max(y.abs().max().item(), 1e-12)
        assert close(da, a.grad), (hidden, n)
        for (dz, ha, hb, lin) in ((dq, h2, None, critic.l3), (dz2, h1, None, critic.l2), (dz1, s, a.detach(), critic.l1)):
            gW, gb = torch.zeros_like(lin.weight), torch.zeros_like(lin.bias)
            mlp.weight_grad(dz, ha, hb, gW, gb)
            assert close(gW, lin.weight.grad) and close(gb, lin.bias.grad), (hidden, n, tuple(gW.shape))
        # actor: dLoss/da given, through a = 0.8 sigmoid(z)
        pa = actor(s)
        g = torch.randn(n, 4, device=dev) / n
        (pa * g).sum().backward()
        la = mlp.layers_of(actor)
        mlp.mlp3_forward(la, s, act=mlp.ACT_SIGMOID, scale=0.8, h1_out=h1, h2_out=h2, shadow=True)
        dz3 = g * pa.detach() * (1 - pa.detach() / 0.8)
        dz2, dz1, _ = mlp.mlp3_backward(la, dz3.contiguous(), h1, h2)
        for (dz, ha, lin) in ((dz3.contiguous(), h2, actor.l3), (dz2, h1, actor.l2), (dz1, s, actor.l1)):
            gW, gb = torch.zeros_like(lin.weight), torch.zeros_like(lin.bias)
            mlp.weight_grad(dz, ha, None, gW, gb)
            assert close(gW, lin.weight.grad) and close(gb, lin.bias.grad), (hidden, n, tuple(gW.shape))
        # the fused sigmoid epilogue of dx: dz3 of an actor whose output pa feeds the critic's action columns
        k1, k2 = torch.empty_like(h1), torch.empty_like(h2)
        mlp.mlp3_forward(lc, s, pa.detach(), h1_out=k1, h2_out=k2, shadow=True)
        _, _, dz3_fused = mlp.mlp3_backward(lc, dq, k1, k2, want_dz=False, dx_cols=(82, 4), act_out=pa.detach().contiguous(), scale=0.8)
        pa2 = pa.detach().clone().requires_grad_(True)
        (critic(s, pa2) * dq).sum().backward()
        assert close(dz3_fused, pa2.grad * pa.detach() * (1 - pa.detach() / 0.8)), (hidden, n)


def test_in_step_rays_equal_the_separate_ray_kernel(monkeypatch):
    """The rays cast at the end of the stepping kernel (wg_rays: two-pass cull, persistent-lane walks, LDS atomicMin) against
    the standalone k_rays launch (KS_RAYS_IN_STEP=0): the nearest hit is a minimum over the same triangle hits - the two
    code paths may contract multiply-adds differently, nothing more - so over a whole episode with auto-resets, for a
    batch that does not fill its last workgroup, the 17 ray distances agree to 1e-5 (all but 0.1 % of them to 5e-7) and the 82-d observations to 2e-5, the
    same rays hit / miss, and rewards and termination flags are identical."""
    n = 1000
    q0, hq = scenarios.config2_states(n)
    acts = torch.as_tensor(scenarios.config_actions(n, 32)).cuda()
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("KS_RAYS_IN_STEP", flag)
        sim = _sim(n, "CubeS", horizon=30, auto_reset=True)
        obs = [sim.reset(torch.as_tensor(q0), torch.as_tensor(hq)).clone()]
        rew, done = [], []
        for t in range(32):
            o, r, d, info = sim.step(acts[t])
            obs.append(o.clone()); rew.append(r.clone()); done.append(d.clone())
        torch.cuda.synchronize()
        outs.append((torch.stack(obs), torch.stack(rew), torch.stack(done), sim.final_obs.clone()))
        sim.close()
    (oa, ra, da, fa), (ob, rb, db, fb) = outs
    assert torch.equal(ra, rb) and torch.equal(da, db)
    assert torch.equal(oa[:, :, 50:67] < 6, ob[:, :, 50:67] < 6)                # the same rays hit something (a miss reads 6)
    print("max |obs difference|", (oa - ob).abs().max().item(), "ray slots", (oa[:, :, 50:67] - ob[:, :, 50:67]).abs().max().item())
    dr = (oa[:, :, 50:67] - ob[:, :, 50:67]).abs()
    print("ray entries differing by > 5e-7:", int((dr > 5e-7).sum()), "of", dr.numel())
    assert dr.max().item() < 1e-5 and (dr > 5e-7).float().mean().item() < 1e-3   # the ray distances themselves (grazing hits amplify rounding)
    assert (oa - ob).abs().max().item() < 2e-5 and (fa - fb).abs().max().item() < 2e-5   # and what the observation derives from them
    assert (oa[:, :, 50:67] < 6).any() and (da != 0).any()                      # rays hit something; episodes ended


def test_full_size_batch_equals_its_slices():
    """BASELINE's full size (4096 envs on one GPU, config 2 rows / actions) against a size-independent property: an env's
    trajectory does not depend on what else is in the batch - envs 512..767 of the 4096-env context walk, bit for bit, the
    trajectory of the same 256 envs alone in a context of their own (other workgroups, another ray-pool population, same kernels)."""
    n, lo, m = 4096, 512, 256
    q0, hq = scenarios.config2_states(n)
    acts = scenarios.config_actions(n, 6)
    big = _sim(n, "CubeS", solver_iterations=SOLVER_ITERATIONS, auto_reset=True, horizon=4)
    small = _sim(m, "CubeS", solver_iterations=SOLVER_ITERATIONS, auto_reset=True, horizon=4)
    ob = big.reset(torch.as_tensor(q0), torch.as_tensor(hq))
    os_ = small.reset(torch.as_tensor(q0[:, lo:lo + m]), torch.as_tensor(hq[:, lo:lo + m]))
    assert torch.equal(ob[lo:lo + m], os_)
    for t in range(6):
        rb = big.step(torch.as_tensor(acts[t]))
        rs = small.step(torch.as_tensor(np.ascontiguousarray(acts[t][:, lo:lo + m])))
        torch.cuda.synchronize()
        assert torch.equal(rb[0][lo:lo + m], rs[0]) and torch.equal(rb[1][lo:lo + m], rs[1]) and torch.equal(rb[2][lo:lo + m], rs[2]), t
    sb, ss = big.get_state(), small.get_state()
    for k in ("qpos", "qvel", "qacc_warmstart"):
        assert torch.equal(sb[k][:, lo:lo + m], ss[k])
    assert int(sb["status"].abs().sum()) == 0
    big.close(); small.close()
