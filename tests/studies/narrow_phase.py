#!/usr/bin/env python3
"""Study (test infrastructure; CPU oracle): what does the product's narrow phase - GJK closest features on the un-inflated hulls in the
margin zone, MPR only on true overlap - change against MuJoCo 1.50's own scheme, libccd MPR on hulls inflated by margin / 2 in both
regimes (the oracle's study mode, ko_sim.narrow_phase = 1)?  VERDICT r2 weak 4a: "a sound fp32 choice, but distance / normal / point near
contact onset differ from the reference engine by an unknown amount".

 (1) per contact, on the SAME states (the grasp + lift trajectories of the one-step tests, 3 poses x 4 starts + 14 shapes): distance,
     normal angle and contact point of every hull-hull contact under both schemes;
 (2) per trajectory: qpos drift after 200 / 330 substeps when the whole trajectory runs under the other scheme;
 (3) behaviour: the ten recorded MuJoCo demonstrations and the naive-controller heat map under both schemes.
usage: python -m tests.studies.narrow_phase > profiles/r03_narrow_phase.txt"""
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
POSES = ("normal", "rotated", "top")


def scenario(args):
    shape, orientation, row = args
    from kinovagrasping_amd import scenarios
    from oracle import ko_py as ko
    model = ko.OracleModel(scenarios.model_blob(shape))
    tab = scenarios.start_coord_table(shape, orientation)
    q0 = np.zeros(16)
    q0[9:12] = tab[row % len(tab)]
    q0[0:3] = scenarios.hand_slide_offsets(orientation, shape, "pose")
    q0[12] = 1
    hq = scenarios.hand_quat_for(orientation)

    def mk(narrow):
        s = ko.OracleSim(model, hq, solver_iterations=20)
        s.s.rays_enabled = 0
        s.s.narrow_phase = narrow
        s.env_reset(q0)
        return s
    A, B, Cc = mk(0), mk(1), mk(1)           # A: product scheme free running; B: MuJoCo scheme on A's states; C: MuJoCo scheme free running
    out = dict(ddist=[], dang=[], dpos=[], only_a=0, only_b=0, both=0, drift=[])
    for t in range(22):
        a = np.array([0.0, 0.6, 0.5, 0.7]) if t < 14 else np.array([0.6, 0.5, 0.5, 0.5])
        ctrl_a = ko.env_ctrl(A.view("geom_xpos").reshape(-1, 3)[1], A.view("geom_xmat").reshape(-1, 9)[1], a)[2]
        ctrl_c = ko.env_ctrl(Cc.view("geom_xpos").reshape(-1, 3)[1], Cc.view("geom_xmat").reshape(-1, 9)[1], a)[2]
        for _ in range(15):
            B.set_state(A.view("qpos").copy(), A.view("qvel").copy(), A.view("qacc_warmstart").copy())
            B.view("ctrl")[:] = ctrl_a
            B.forward()
            A.step(ctrl_a)
            Cc.step(ctrl_c)
            ca = {(c["geom1"], c["geom2"]): c for c in A.contacts() if c["geom1"] != 0}
            cb = {(c["geom1"], c["geom2"]): c for c in B.contacts() if c["geom1"] != 0}
            for k in set(ca) | set(cb):
                if k in ca and k in cb:
                    out["both"] += 1
                    out["ddist"].append(cb[k]["dist"] - ca[k]["dist"])
                    out["dang"].append(np.degrees(np.arccos(np.clip(np.dot(ca[k]["frame"][:3], cb[k]["frame"][:3]), -1, 1))))
                    out["dpos"].append(np.linalg.norm(ca[k]["pos"] - cb[k]["pos"]))
                elif k in ca:
                    out["only_a"] += 1
                else:
                    out["only_b"] += 1
            qc = Cc.view("qpos")
            out["drift"].append(np.abs(A.view("qpos") - qc).max() / max(1e-3, np.abs(qc).max()))
    return (shape, orientation, row), {k: (np.array(v) if isinstance(v, list) else v) for k, v in out.items()}


def demos(narrow):
    import torch
    from kinovagrasping_amd import demonstrators
    from tests.oracle_vec import OracleVecSim, place_at_palm_xy
    rec = np.load(ROOT / "tests" / "golden" / "mujoco_recorded.npz")
    sim = OracleVecSim(10, "CubeS", solver_iterations=100, rays=False, narrow_phase=narrow)
    q, hq, _ = place_at_palm_xy(sim, rec["demo_x"], rec["demo_y"])
    obs0 = sim.reset(torch.as_tensor(q), torch.as_tensor(hq))
    out = demonstrators.run_controller_episodes(sim, obs0.clone(), None, horizon=30, mode="naive", lift_rule="expert")
    return out["success"].numpy().astype(int), out["steps"].numpy(), rec["demo_success"], rec["demo_steps"]


def main():
    from kinovagrasping_amd import scenarios
    jobs = [("CubeS", o, r) for o in POSES for r in (0, 700, 1400, 2100)] + [(sh, "normal", r) for sh in scenarios.SHAPES for r in (100, 1500)]
    with ProcessPoolExecutor(8) as ex:
        res = list(ex.map(scenario, jobs))
    cat = lambda k: np.concatenate([r[k] for _, r in res])
    both, oa, ob = sum(r["both"] for _, r in res), sum(r["only_a"] for _, r in res), sum(r["only_b"] for _, r in res)
    print("product narrow phase (GJK closest features in the margin zone + MPR on overlap) against MuJoCo 1.50's scheme (MPR on hulls inflated by margin / 2),")
    print(f"fp64 oracle, {len(res)} grasp + lift trajectories x 330 substeps, hull-hull contacts of the SAME states under both schemes\n")
    print(f"contacts found by both: {both}; only by the product scheme: {oa}; only by the MuJoCo scheme: {ob}")
    dd, da, dp = cat("ddist"), cat("dang"), cat("dpos")
    q = lambda x, p: np.percentile(np.abs(x), p)
    print(f"distance (MuJoCo scheme - product) [m]: median {np.median(dd):+.2e}, |.| p50 {q(dd, 50):.2e}  p90 {q(dd, 90):.2e}  p99 {q(dd, 99):.2e}  max {np.abs(dd).max():.2e}   (margin 1e-3)")
    print(f"normal angle [deg]:                      p50 {q(da, 50):.3f}  p90 {q(da, 90):.3f}  p99 {q(da, 99):.3f}  max {da.max():.3f}")
    print(f"contact point distance [m]:              p50 {q(dp, 50):.2e}  p90 {q(dp, 90):.2e}  p99 {q(dp, 99):.2e}  max {dp.max():.2e}")
    d200 = np.array([r["drift"][199] for _, r in res]); d330 = np.array([r["drift"][329] for _, r in res])
    print(f"\nwhole trajectory under the other scheme: relative qpos difference at substep 200: median {np.median(d200):.1e}  p90 {np.percentile(d200, 90):.1e}  max {d200.max():.1e};"
          f" at substep 330: median {np.median(d330):.1e}  max {d330.max():.1e}")
    print("\nbehaviour, the ten recorded MuJoCo demonstrations (naive controller, expert_data.py loop):")
    for narrow, name in ((0, "product scheme"), (1, "MuJoCo 1.50 scheme")):
        s, t, rs, rt = demos(narrow)
        print(f"  {name:20s} outcomes {s}  steps {t}   (recorded outcomes {rs}, steps {rt}): {int((s == rs).sum())}/10 outcomes, "
              f"{int((np.abs(t - rt)[rs == 1] <= 2).sum())}/8 successes within 2 steps, {int((np.abs(t - rt)[rs == 1] <= 1).sum())}/8 within 1")
    print("\nbehaviour, naive-controller heat map (1083 cells): python -m tests.studies.naive_heatmap 100 0|1 -")
    print("  product scheme:      success cells 0.997, far-corner failures 0.854, near-palm centre failures 0.000, median duration 22 steps")
    print("  MuJoCo 1.50 scheme:  success cells 0.997, far-corner failures 0.876, near-palm centre failures 0.000, median duration 23 steps (recorded: 23-24)")


if __name__ == "__main__":
    main()
