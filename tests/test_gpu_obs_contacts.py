"""GPU parity, second set: what the HIP path RETURNS after stepping (observations, qvel, per-contact geometry and
forces) against the fp64 CPU oracle, in all three hand poses (normal / rotated / top), plus the reference-generated
golden vectors pushed through the HIP kernels themselves (env layer -> ks_obs_from_snapshot, learner -> the native
DDPGfD update).  Run with `-m gpu` on an MI355X; everything goes through the C ABI (libkinova_sim.so).

Tolerances, stated once (fp32 product vs fp64 oracle unless noted):
  * teacher-forced single mj_step, states whose contact POINTS agree with the oracle's (every state but the ties on
    parallel features, DESIGN.md "known limits"): |dqpos| <= 5e-6, |dqvel| <= 5e-4 (= dqpos / dt), contact distance
    <= 2e-6, normal <= 2e-4, contact force <= 2e-3 relative to the largest force of the state;  a state whose
    contact point differs must still have the same pair list, distance and normal (that is what makes it a tie).
  * fp64 instantiation of the same kernels: qpos 1e-9, qvel 1e-7, forces 1e-6 relative.
  * observation after one env-step from the oracle's own state (15 substeps, rays, 82 slots): 2e-4 absolute + relative,
    x50 on the slots SURVEY O2 flags as ill conditioned (48-49 angles, 73-74 area ratios, 75-81 twentieth powers);
    ray slots (50-66, 70-72) may differ where a ray grazes an edge: at most 1 % of them beyond the tolerance.
"""
import ctypes
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import ko_py as ko
from kinovagrasping_amd import scenarios
from kinovagrasping_amd.sim import SOLVER_ITERATIONS

pytestmark = pytest.mark.gpu
POSES = ("normal", "rotated", "top")
GOLDEN = Path(__file__).resolve().parent / "golden"


def _sim(*a, **k):
    from kinovagrasping_amd.sim import KinovaSim
    return KinovaSim(*a, **k)


def obs_tolerance(base):
    t = np.full(82, base)
    t[48:50] *= 50
    t[73:82] *= 50
    return t


RAY_SLOTS = np.r_[50:67, 70:73]
NON_RAY = np.setdiff1d(np.arange(82), RAY_SLOTS)


def pose_start(shape, orientation, row):
    tab = scenarios.start_coord_table(shape, orientation)
    q0 = np.zeros(16)
    q0[9:12] = tab[row % len(tab)]
    q0[12] = 1
    return q0, scenarios.hand_quat_for(orientation)


def ctrl_of(o, action):
    """the 9 controls the env layer derives from a 4-d action at the oracle's current palm pose (ENV:1495-1535)"""
    return ko.env_ctrl(o.view("geom_xpos").reshape(-1, 3)[1], o.view("geom_xmat").reshape(-1, 9)[1], action)[2]


def oracle_substep_records(model, shape, orientation, row, n_env_steps=22):
    """closing grasp then lift (the reference's scripted lift action), every substep recorded with its contacts"""
    q0, hq = pose_start(shape, orientation, row)
    o = ko.OracleSim(model, hq, solver_iterations=SOLVER_ITERATIONS)
    o.env_reset(q0)
    rec = []
    for t in range(n_env_steps):
        a = np.array([0.0, 0.6, 0.5, 0.7]) if t < 14 else np.array([0.6, 0.5, 0.5, 0.5])
        ctrl = ctrl_of(o, a)
        for _ in range(15):
            before = (o.view("qpos").copy(), o.view("qvel").copy(), o.view("qacc_warmstart").copy())
            o.step(ctrl)
            cs = o.contacts()
            rec.append(dict(before=before, ctrl=ctrl.copy(), qpos=o.view("qpos").copy(), qvel=o.view("qvel").copy(), ncon=o.s.ncon,
                            pos=np.array([c["pos"] for c in cs]).reshape(-1, 3), normal=np.array([c["frame"][:3] for c in cs]).reshape(-1, 3),
                            dist=np.array([c["dist"] for c in cs]), force=o.contact_forces(),
                            bodies=np.array([ko.GEOM_BODY[c["geom1"]] + 16 * ko.GEOM_BODY[c["geom2"]] for c in cs], dtype=int)))
    return hq, rec


def gpu_one_substep(precision, shape, hq, rec):
    n = len(rec)
    sim = _sim(n, shape, precision=precision, solver_iterations=SOLVER_ITERATIONS, contact_tap=True)
    q0 = np.stack([r["before"][0] for r in rec], 1)
    sim.reset(torch.as_tensor(q0), torch.as_tensor(np.repeat(hq[:, None], n, 1)))
    sim.set_state(torch.as_tensor(q0), torch.as_tensor(np.stack([r["before"][1] for r in rec], 1)),
                  torch.as_tensor(np.stack([r["before"][2] for r in rec], 1)))
    sim.substep(torch.as_tensor(np.stack([r["ctrl"] for r in rec], 1)))
    st = sim.get_state(contacts=True)
    torch.cuda.synchronize()
    out = {k: v.double().cpu().numpy() if v.is_floating_point() else v.cpu().numpy() for k, v in st.items()}
    sim.close()
    return out


@pytest.mark.parametrize("orientation", POSES)
def test_one_step_qvel_and_contact_forces_in_every_pose(orientation):
    """north_star: 'per-step qpos / qvel / contact-force trajectories match'.  330 states of a CubeS grasp + lift per pose,
    each advanced by ONE mj_step on the GPU from the oracle's state: qpos, qvel, contact list (pairs, distance, normal,
    point) and per-contact force (normal + two tangents, from the contact tap) against the oracle's."""
    model = ko.OracleModel(scenarios.model_blob("CubeS"))
    hq, rec = oracle_substep_records(model, "CubeS", orientation, row=1234)
    n = len(rec)
    assert max(r["ncon"] for r in rec) >= (5 if orientation == "normal" else 10)
    for precision in (64, 32):
        g = gpu_one_substep(precision, "CubeS", hq, rec)
        assert (g["status"] & 2 == 0).all()
        f64 = precision == 64
        eq = np.array([np.abs(g["qpos"][:, i] - rec[i]["qpos"]).max() for i in range(n)])
        ev = np.array([np.abs(g["qvel"][:, i] - rec[i]["qvel"]).max() for i in range(n)])
        same_n = np.array([g["ncon"][i] == rec[i]["ncon"] for i in range(n)])
        agree, tie, e_dist, e_nrm, e_force, nforce, odd = np.zeros(n, bool), np.zeros(n, bool), [], [], [], 0, []
        for i in range(n):
            r, nc = rec[i], rec[i]["ncon"]
            if not same_n[i]:
                continue
            c = g["contact"][:nc, :, i]
            if nc == 0:
                agree[i] = True
                continue
            same_pairs = ((c[:, 8].astype(int) & 255) == r["bodies"]).all()
            dd, dn = np.abs(c[:, 6] - r["dist"]).max(), np.abs(c[:, 3:6] - r["normal"]).max()
            dp = np.abs(c[:, 0:3] - r["pos"]).max()
            geometry_ok = same_pairs and dd <= (1e-9 if f64 else 2e-6) and dn <= (1e-7 if f64 else 5e-4)
            agree[i] = geometry_ok and dp <= (1e-7 if f64 else 1e-4)
            tie[i] = geometry_ok and not agree[i]
            if not geometry_ok:
                odd.append((i, nc, bool(same_pairs), float(dd), float(dn), float(dp), float(eq[i])))
            if agree[i]:
                e_dist.append(dd); e_nrm.append(dn)
                scale = max(1e-3, np.abs(r["force"]).max())
                e_force.append(np.abs(c[:, 14:17] - r["force"]).max() / scale)
                nforce += int((r["force"][:, 0] > 0).sum())
        e_force = np.array(e_force)
        print(f"{orientation} fp{precision}: {n} states, contact counts equal {same_n.mean():.3f}, contact points agree {agree.mean():.3f}, "
              f"ties {tie.mean():.3f}; agreeing states: |dqpos| max {eq[agree].max():.2e}, |dqvel| max {ev[agree].max():.2e}, "
              f"force err max {e_force.max():.2e} (median {np.median(e_force):.2e}) over {nforce} loaded contacts; "
              f"all states: |dqpos| median {np.median(eq):.2e} max {eq.max():.2e}, |dqvel| max {ev.max():.2e}")
        for row in odd[:40]:
            print("   state %d ncon %d same pairs %s: d dist %.2e d normal %.2e d point %.2e -> |dqpos| %.2e" % row)
        assert nforce >= 200
        if f64:
            assert same_n.all()
            assert (agree | tie).all()                           # every state is either exact or an explained tie
            assert eq[agree].max() <= 1e-9 and ev[agree].max() <= 1e-7 and e_force.max() <= 1e-6
            assert agree.mean() >= 0.97
        else:
            assert same_n.mean() >= 0.98                         # a contact at 1e-7 of its margin may flip in fp32
            # (near-touching pairs: the closest-feature normal -v / |v| loses its conditioning as |v| -> 0 and the query hands
            # over to the penetration query at a sign decided by rounding: a few states per 330 with normals 0.2 degrees apart)
            # (round 4: every object contact is a penetration contact from MPR - explicit pairs have margin 0 - and 2 % of the states of
            # the 'normal' pose have a finger pad flat on a cube face, where the fp32 query ends on another triangle of that face)
            # (measured at the end of round 5: counts equal 1.000, points agree 0.994 / 0.994 / 1.000, agreeing states |dqpos| <= 1.0e-6, |dqvel| <= 1.0e-4,
            # force error <= 4.7e-4; the worst state of all 1.4e-4)
            assert (agree | tie)[same_n].mean() >= 0.98
            assert eq[agree].max() <= 3e-6 and ev[agree].max() <= 3e-4
            assert e_force.max() <= 1.5e-3
            assert agree.mean() >= 0.98
        # the tail (ties + flipped counts) stays bounded: a contact point moved along a flat feature, nothing else
        assert eq.max() <= (5e-3 if precision == 64 else 1.5e-3)


@pytest.mark.parametrize("orientation", POSES)
def test_post_step_observation_reward_done_from_oracle_states(orientation):
    """The product observation path (k_env_step's in-kernel rays -> k_obs) AFTER stepping, in every pose: 96 envs (4 start
    rows x 24 points of a random-action episode) are put into the oracle's state at that point (ks_set_state) and take ONE
    env.step with the same action; the 82-d observation, reward, done and qpos / qvel are compared with the oracle's."""
    model = ko.OracleModel(scenarios.model_blob("CubeS"))
    states, acts, ref = [], [], []
    for row in (0, 700, 2100, 3900):
        q0, hq = pose_start("CubeS", orientation, row)
        o = ko.OracleSim(model, hq, solver_iterations=SOLVER_ITERATIONS)
        o.env_reset(q0)
        a = scenarios.config_actions(1, 24, base_seed=40 + row)[:, :, 0].astype(np.float64)
        a[:, 1:] = np.abs(a[:, 1:])                               # fingers close, wrist wanders: contact-rich states
        for t in range(24):
            states.append((q0, o.view("qpos").copy(), o.view("qvel").copy(), o.view("qacc_warmstart").copy()))
            ob, r, d, info = o.env_step(a[t])
            acts.append(a[t]); ref.append((ob, r, d, o.view("qpos").copy(), o.view("qvel").copy()))
    n = len(states)
    hqn = np.repeat(hq[:, None], n, 1)
    for precision, tol in ((64, 1e-8), (32, 2e-4)):
        sim = _sim(n, "CubeS", precision=precision, horizon=0)
        sim.reset(torch.as_tensor(np.stack([s[0] for s in states], 1)), torch.as_tensor(hqn))
        sim.set_state(*(torch.as_tensor(np.stack([s[k] for s in states], 1)) for k in (1, 2, 3)))
        og, rg, dg, ig = sim.step(torch.as_tensor(np.stack(acts, 1)))
        torch.cuda.synchronize()
        st = sim.get_state()
        og = og.double().cpu().numpy()
        oo = np.stack([r[0] for r in ref])
        t_obs = obs_tolerance(tol)
        err = np.abs(og - oo)
        bad = err > t_obs + t_obs * np.abs(oo)
        eq = np.abs(st["qpos"].double().cpu().numpy() - np.stack([r[3] for r in ref], 1)).max(0)
        ev = np.abs(st["qvel"].double().cpu().numpy() - np.stack([r[4] for r in ref], 1)).max(0)
        # envs whose 15 substeps stayed on the oracle's trajectory (no contact flipped): everything must agree
        on = eq <= (1e-8 if precision == 64 else 2e-5)
        print(f"{orientation} fp{precision}: {n} env-steps, on-trajectory {on.mean():.3f}; obs slots beyond tolerance "
              f"(non-ray) {int(bad[on][:, NON_RAY].sum())}, (ray) {int(bad[on][:, RAY_SLOTS].sum())} of {int(on.sum()) * len(RAY_SLOTS)}; "
              f"|dqpos| median {np.median(eq):.2e} max {eq.max():.2e}; |dqvel| median {np.median(ev):.2e}")
        assert on.mean() >= (0.95 if precision == 64 else 0.93)                 # (fp32 measured 0.969 / 1.000 / 1.000)
        assert not bad[on][:, NON_RAY].any(), np.argwhere(bad[on][:, NON_RAY])[:5]
        assert bad[on][:, RAY_SLOTS].mean() <= 0.01
        assert (ev[on] <= (1e-6 if precision == 64 else 2e-3)).all()
        assert (rg.double().cpu().numpy()[on] == np.array([r[1] for r in ref])[on]).all()
        assert ((dg.cpu().numpy() & 1).astype(bool)[on] == np.array([r[2] for r in ref])[on]).all()
        assert (st["status"].cpu().numpy() & 2 == 0).all()
        sim.close()


@pytest.mark.parametrize("orientation", POSES)
def test_free_running_episode_observations_in_every_pose(orientation):
    """BASELINE config 1 in each pose: PCG64(0) actions, 30 env-steps free running, fp32 GPU against the oracle.  Every
    observation up to the first env-step whose qpos leaves the 1e-4 relative band is asserted (not only the reset one);
    the band must hold for at least 200 substeps (north_star)."""
    model = ko.OracleModel(scenarios.model_blob("CubeS"))
    q0, hq = pose_start("CubeS", orientation, 0)
    acts = scenarios.config_actions(1, 30, base_seed=0)[:, :, 0]
    o = ko.OracleSim(model, hq, solver_iterations=SOLVER_ITERATIONS)
    sim = _sim(1, "CubeS", solver_iterations=SOLVER_ITERATIONS, horizon=0)
    ob0 = o.env_reset(q0)
    og0 = sim.reset(torch.as_tensor(q0[:, None]), torch.as_tensor(hq[:, None])).double().cpu().numpy()[0].copy()
    np.testing.assert_allclose(og0, ob0, rtol=2e-4, atol=2e-5)
    first_bad, checked, worst = None, 0, 0.0
    for t in range(30):
        ob, r, d, info = o.env_step(acts[t])
        og, rg, dg, ig = sim.step(torch.as_tensor(acts[t][:, None]))
        torch.cuda.synchronize()
        st = sim.get_state()
        qg, qo = st["qpos"].double().cpu().numpy()[:, 0], o.view("qpos")
        rel = np.abs(qg - qo).max() / max(1e-3, np.abs(qo).max())
        if first_bad is None and rel > 1e-4:
            first_bad = (t, rel)
        if first_bad is None:
            og = og.double().cpu().numpy()[0]
            t_obs = obs_tolerance(1e-3)                         # free running: 15 x (t + 1) substeps of fp32 drift behind it
            bad = np.abs(og - ob) > t_obs + t_obs * np.abs(ob)
            assert not bad[NON_RAY].any(), (t, np.argwhere(bad).ravel(), og[bad], ob[bad])
            assert bad[RAY_SLOTS].sum() <= 1, (t, np.argwhere(bad).ravel())
            ev = np.abs(st["qvel"].double().cpu().numpy()[:, 0] - o.view("qvel")).max()
            worst = max(worst, ev)
            assert ev <= 2e-2, (t, ev)
            assert rg.item() == r and bool(dg.item() & 1) == d
            checked += 1
    print(f"{orientation}: observations asserted for {checked} env-steps; first env-step beyond 1e-4 relative qpos error: {first_bad}; "
          f"worst |dqvel| while on trajectory {worst:.2e}")
    assert first_bad is None or first_bad[0] * 15 >= 200, first_bad
    sim.close()


def test_env_layer_golden_vectors_through_the_hip_observation_kernel():
    """tests/golden/env_layer.npz holds what the REFERENCE's own _get_obs / _get_reward returned (tests/golden/gen_golden_env.py
    ran kinova_gripper_env.py) for 72 engine states x 4 shapes x 3 hand poses, both palm-sensor branches, the lift
    threshold straddled.  The same engine states (body poses + the 26 sensor values) go through the HIP observation
    kernel (ks_obs_from_snapshot -> k_obs -> build_obs) in both precisions."""
    G = np.load(GOLDEN / "env_layer.npz")
    shapes = [str(s) for s in G["shapes"]]
    total = 0
    for si, shape in enumerate(shapes):
        idx = np.flatnonzero(G["shape_idx"].astype(int) == si)
        n = len(idx)
        snap = np.zeros((105, n))
        for b in range(2, 10):
            snap[(b - 2) * 12:(b - 2) * 12 + 9] = G["body_xmat"][idx, b].T
            snap[(b - 2) * 12 + 9:(b - 2) * 12 + 12] = G["body_xpos"][idx, b].T
        snap[96:105] = G["sensordata"][idx, :9].T
        rays = G["sensordata"][idx, 9:].T
        for precision, tol in ((64, 1e-9), (32, 2e-5)):
            sim = _sim(n, shape, precision=precision)
            og, rg, dg, ig = sim.obs_from_snapshot(torch.as_tensor(snap), torch.as_tensor(rays))
            torch.cuda.synchronize()
            og = og.double().cpu().numpy()
            ref = G["obs_local"][idx]
            t_obs = obs_tolerance(tol)
            err = np.abs(og - ref)
            assert (err <= t_obs + t_obs * np.abs(ref)).all(), (shape, precision, np.argwhere(err > t_obs + t_obs * np.abs(ref))[:5], err.max())
            np.testing.assert_array_equal(rg.double().cpu().numpy(), G["reward"][idx])
            np.testing.assert_array_equal((dg.cpu().numpy() & 1).astype(bool), G["done"][idx].astype(bool))
            np.testing.assert_array_equal(ig.double().cpu().numpy().T, G["info"][idx])
            sim.close()
        total += n
    hit = G["palm_hit"].astype(bool)
    assert total == 72 and hit.any() and (~hit).any() and G["done"].any() and not G["done"].all()


def test_learner_golden_vectors_through_the_native_update():
    """tests/golden/learner.npz holds the REFERENCE's DDPGfD.train_batch losses and parameters (tools/gen_golden_learner.py
    ran DDPGfD.py / utils.py, 400-300 widths, 10 calls on its own seeded sample stream).  Here the same initial weights
    and the same sampled batches go through learner_native.NativeDDPGfDUpdate on the GPU (fused MFMA forwards, explicit
    backward GEMMs, kr_critic_grad / kr_adam_step / kr_soft_update): losses to 1e-5 relative, parameters to 1e-6
    absolute after 1 and 10 calls (10 crosses the soft target update)."""
    from kinovagrasping_amd.ddpgfd import DDPGfD
    from kinovagrasping_amd.learner_native import NativeDDPGfDUpdate
    from tests.test_learner_golden import build_buffers, load_init
    G = np.load(GOLDEN / "learner.npz")
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    pol = DDPGfD(82, 4, 0.8, 5, batch_size=6, device=dev)
    load_init(pol, G)
    nat = NativeDDPGfDUpdate(pol)
    nat.track_actor_loss = True
    agent, expert = build_buffers(G)
    np.random.seed(5)
    for it in range(10):
        ag = agent.sample_batch_nstep(int(6 * 0.7))
        ex = expert.sample_batch_nstep(6 - int(6 * 0.7))
        batch = [torch.cat((a, e), 0) for a, e in zip(ag, ex)]
        if it == 0:
            for j, nm in enumerate(("state", "action", "next", "reward", "not_done")):
                np.testing.assert_array_equal(batch[j].numpy(), G[f"batch0_{nm}"])
        st, ac, ns, rw = (t.to(dev) for t in batch[:4])
        lc = nat.train_on_batch(st, ac, ns, rw)
        losses = [nat.actor_loss.item()] + [x.item() for x in lc]
        # critic losses: 1e-5 relative.  The actor loss is the mean of Q values of magnitude ~5 and mixed sign (the golden
        # critic loss is ~20), so its own magnitude (0.06) is no scale for rounding: 1e-5 absolute (= 2e-6 of |Q|)
        np.testing.assert_allclose(losses[1:], G["losses"][it][1:], rtol=1e-5, atol=1e-6, err_msg=f"call {it}")
        np.testing.assert_allclose(losses[0], G["losses"][it][0], rtol=1e-5, atol=1e-5, err_msg=f"call {it} (actor loss)")
        if it in (0, 9):
            for name, net in (("actor", pol.actor), ("critic", pol.critic), ("actor_target", pol.actor_target), ("critic_target", pol.critic_target)):
                for k, v in net.state_dict().items():
                    v = v.cpu().numpy()
                    np.testing.assert_allclose(v.ravel()[::max(1, v.size // 256)][:256], G[f"after{it + 1}_{name}.{k}.sample"], rtol=0, atol=1e-6,
                                               err_msg=f"{name}.{k} after {it + 1}")
                    stats = G[f"after{it + 1}_{name}.{k}.stats"]
                    np.testing.assert_allclose([v.astype(np.float64).sum(), np.abs(v.astype(np.float64)).sum()], stats, rtol=1e-6, atol=1e-4)


def test_fourteen_shapes_three_poses_observations_track_the_oracle():
    """Every README object x every hand pose x 4 starts: 4 env-steps of a closing grasp on the GPU against the oracle -
    qpos AND the returned observations (BASELINE config 5's state space at nominal mass / friction).

    The reference's 'rotated' / 'top' starts put the hand into the floor and, for the larger objects, the object into the
    hand (SURVEY note N5: the training driver builds a fresh env per episode, so determine_hand_location multiplies its
    offsets by a zero Tfw, main_DDPGfD.py:381 + ENV:110,1299): the constraint solver then ejects the object at metres
    per second.  Such LAUNCHED envs (oracle object speed > 1 m/s at any env-step) are still exact in the fp64 kernels; in
    fp32 a ballistic, tumbling object is only required to stay finite (its error is reported) - everything else must track."""
    per = 4
    act = np.array([0.0, 0.6, 0.5, 0.7])
    summary, launched_total = [], 0
    for sh in scenarios.SHAPES:
        model = ko.OracleModel(scenarios.model_blob(sh))
        q0s, hqs = [], []
        for ori in POSES:
            for k in range(per):
                q0, hq = pose_start(sh, ori, 17 + 997 * k)
                q0s.append(q0); hqs.append(hq)
        n = len(q0s)
        qg, og = {}, {}
        for prec in (32, 64):
            sim = _sim(n, sh, horizon=0, precision=prec)
            sim.reset(torch.as_tensor(np.stack(q0s, 1)), torch.as_tensor(np.stack(hqs, 1)))
            for t in range(4):
                o_ = sim.step(torch.as_tensor(np.repeat(act[:, None], n, 1)))[0]
            torch.cuda.synchronize()
            st = sim.get_state()
            assert (st["status"].cpu().numpy() & 2 == 0).all(), (sh, prec)
            qg[prec], og[prec] = st["qpos"].double().cpu().numpy(), o_.double().cpu().numpy()
            sim.close()
        rel32, rel64, launched, nbad, nray = np.zeros(n), np.zeros(n), np.zeros(n, bool), 0, 0
        t_obs = obs_tolerance(5e-4)
        for i in range(n):
            o = ko.OracleSim(model, hqs[i], solver_iterations=SOLVER_ITERATIONS)
            o.env_reset(q0s[i])
            for t in range(4):
                ob = o.env_step(act)[0]
                launched[i] |= np.abs(o.view("qvel")[9:12]).max() > 1.0
            qo = o.view("qpos")
            rel32[i] = np.abs(qg[32][:, i] - qo).max() / max(1e-3, np.abs(qo).max())
            rel64[i] = np.abs(qg[64][:, i] - qo).max() / max(1e-3, np.abs(qo).max())
            if rel32[i] <= 1e-4:
                bad = np.abs(og[32][i] - ob) > t_obs + t_obs * np.abs(ob)
                nbad += int(bad[NON_RAY].sum()); nray += int(bad[RAY_SLOTS].sum())
        tame = ~launched
        launched_total += int(launched.sum())
        summary.append((sh, float(np.median(rel32[tame])), float(rel32[tame].max()), int((rel32[tame] <= 2e-4).sum()), int(tame.sum()),
                        float(rel32[launched].max()) if launched.any() else 0.0, float(np.sort(rel64)[-3]), nbad, nray))
    for row in summary:
        print("%-10s fp32 tame envs: median rel qpos %.2e max %.2e, %d/%d within 2e-4; launched max %.2e; fp64 third-worst %.1e; "
              "obs slots off: %d non-ray, %d ray" % row)
    print("launched envs:", launched_total, "of", 14 * 12)
    assert all(r[1] <= 5e-5 for r in summary), summary
    assert all(r[3] >= r[4] - 2 for r in summary), summary       # at most two tame grasps per shape flipped a contact
    assert all(np.isfinite(r[5]) for r in summary), summary
    assert all(r[6] <= 1e-9 for r in summary), summary           # fp64 kernels: the logic is the oracle's (two warm-start ties allowed)
    assert all(r[7] == 0 for r in summary), summary
    assert sum(r[8] for r in summary) <= 0.01 * 14 * 12 * len(RAY_SLOTS)
    assert launched_total <= 14 * 12 // 2


@pytest.mark.gpu
def test_step_finished_inside_the_stepping_kernel_equals_the_separate_observation_launch(monkeypatch):
    """ks_step's observation / reward / done / auto-reset are produced at the tail of the stepping kernel (wg_obs); with
    KS_OBS_IN_STEP=0 a context uses the separate k_obs launch (what fp64 contexts and ks_reset use).  Both run the same
    source (obs_epilogue), so an episode with lifts, the time limit and restarts must agree: flags and rewards exactly,
    observations and states to the last bits (different inlining contexts may contract an fma differently)."""
    from kinovagrasping_amd.sim import KinovaSim
    n, horizon = 256, 12
    q0, hq = scenarios.config2_states(n)
    acts = torch.as_tensor(scenarios.config_actions(n, 30)).cuda()
    # half of the envs start with the object above the lift target: done by `lifted` in the first step
    q0 = q0.copy(); q0[11, ::2] = 0.25
    monkeypatch.setenv("KS_OBS_IN_STEP", "0")
    a = KinovaSim(n, "CubeS", auto_reset=True, horizon=horizon)
    monkeypatch.delenv("KS_OBS_IN_STEP")
    b = KinovaSim(n, "CubeS", auto_reset=True, horizon=horizon)
    oa, ob = a.reset(torch.as_tensor(q0), torch.as_tensor(hq)), b.reset(torch.as_tensor(q0), torch.as_tensor(hq))
    assert torch.equal(oa, ob)
    restarts = 0
    for t in range(2 * horizon + 3):
        ra, rb = a.step(acts[t]), b.step(acts[t])
        torch.cuda.synchronize()
        assert torch.equal(ra[2], rb[2]) and torch.equal(ra[1], rb[1]), t            # done, reward
        assert torch.equal(ra[3], rb[3]), t                                           # info
        tol = torch.full((82,), 1e-6, device="cuda"); tol[73:82] = 5e-5          # 73-81: ratios and 20th powers
        assert a.obs.shape == (n, 82)
        close = lambda x, y: bool(((x - y).abs() <= tol + tol * y.abs()).all())
        assert close(ra[0], rb[0]), t
        d = ra[2].bool()
        restarts += int(d.sum())
        if d.any():
            assert close(a.final_obs[d], b.final_obs[d]), t
        sa, sb = a.get_state(), b.get_state()
        for k in ("qpos", "qvel", "qacc_warmstart"):
            torch.testing.assert_close(sa[k], sb[k], rtol=0, atol=0)
    assert restarts >= 2 * n        # lifted envs restart every step, the others at the time limit
    a.close(); b.close()


@pytest.mark.gpu
def test_pooled_rays_equal_the_rays_every_workgroup_casts_for_itself(monkeypatch):
    """At 4096 envs (one workgroup per CU) the stepping launch pools the rangefinder rays: workgroups publish their envs'
    snapshot a substep early and late finishers cast the stragglers' rays (wg_ray_pool).  Who casts a ray must not matter:
    observations, rewards, flags and states are bit-identical to a context with KS_RAY_POOL=0, and no env reports a pool
    time-out."""
    from kinovagrasping_amd.sim import KinovaSim
    n = 4096
    q0, hq = scenarios.config2_states(n)
    acts = torch.as_tensor(scenarios.config_actions(n, 6)).cuda()
    monkeypatch.setenv("KS_RAY_POOL", "0")
    a = KinovaSim(n, "CubeS", auto_reset=True, horizon=4)
    monkeypatch.delenv("KS_RAY_POOL")
    b = KinovaSim(n, "CubeS", auto_reset=True, horizon=4)
    assert torch.equal(a.reset(torch.as_tensor(q0), torch.as_tensor(hq)), b.reset(torch.as_tensor(q0), torch.as_tensor(hq)))
    for t in range(6):
        ra, rb = a.step(acts[t]), b.step(acts[t])
        torch.cuda.synchronize()
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), t
        assert torch.equal(a.final_obs, b.final_obs)
    sa, sb = a.get_state(), b.get_state()
    for k in ("qpos", "qvel", "qacc_warmstart", "ncon"):
        assert torch.equal(sa[k], sb[k])
    assert int(sb["status"].abs().sum()) == 0 and int(sa["status"].abs().sum()) == 0
    a.close(); b.close()


@pytest.mark.gpu
def test_ks_step_replayed_from_a_hip_graph_equals_direct_launches():
    """every call is asynchronous on the caller's stream and does no host synchronisation, so ks_step can be
    captured in a HIP graph (the output record travels through pinned memory, the launch's ray pool cleans up after itself).
    A captured step replayed 8 times must walk the same trajectory as 8 direct calls."""
    from kinovagrasping_amd.sim import KinovaSim
    n = 4096
    q0, hq = scenarios.config2_states(n)
    a, b = KinovaSim(n, "CubeS", auto_reset=True, horizon=5), KinovaSim(n, "CubeS", auto_reset=True, horizon=5)
    a.reset(torch.as_tensor(q0), torch.as_tensor(hq)); b.reset(torch.as_tensor(q0), torch.as_tensor(hq))
    acts = torch.as_tensor(scenarios.config_actions(n, 9)).cuda()
    act = acts[0].clone()
    # warm-up on a side stream (as torch.cuda.graphs requires), then capture ONE step reading the static action buffer
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        a.step(act)
    torch.cuda.current_stream().wait_stream(s)
    b.step(acts[0])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        a.step(act)
    b.step(acts[0])               # the captured call itself is not executed: b takes that step after the capture ...
    act.copy_(acts[0]); g.replay()  # ... and a replays it
    for t in range(1, 9):
        act.copy_(acts[t])
        g.replay()
        rb = b.step(acts[t])
        torch.cuda.synchronize()
        assert torch.equal(a.obs, rb[0]) and torch.equal(a.reward, rb[1]) and torch.equal(a.done, rb[2]), t
    sa, sb = a.get_state(), b.get_state()
    assert torch.equal(sa["qpos"], sb["qpos"]) and int(sa["status"].abs().sum()) == 0
    a.close(); b.close()
