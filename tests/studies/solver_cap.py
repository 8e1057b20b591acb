#!/usr/bin/env python3
"""Study (test infrastructure; CPU oracle only): what does the Newton cap of 6 iterations - which the product kernels AND the
oracle of the parity tests share - cost against MuJoCo's own setting (<= 100 iterations, tolerance 1e-8)?

For every scenario a closing grasp + scripted lift (the trajectories of tests/test_gpu_obs_contacts.py: 22 env-steps = 330
substeps) is run three ways:
  A  free running, cap 6, step tolerance 1e-5          (ks_config.solver_iterations = 6: the product setting)
  B  at EVERY substep of A, from A's state: cap 100, step tolerance 1e-10  -> the converged qacc of that very problem
  C  free running, cap 100, tolerance 1e-10             -> trajectory drift of A
usage: python -m tests.studies.solver_cap > profiles/r03_solver_cap.txt"""
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
POSES = ("normal", "rotated", "top")


def scenario(args):
    shape, orientation, row, cap = args
    from kinovagrasping_amd import scenarios
    from oracle import ko_py as ko
    model = ko.OracleModel(scenarios.model_blob(shape))
    tab = scenarios.start_coord_table(shape, orientation)
    q0 = np.zeros(16)
    q0[9:12] = tab[row % len(tab)]
    q0[0:3] = scenarios.hand_slide_offsets(orientation, shape, "pose")
    q0[12] = 1
    hq = scenarios.hand_quat_for(orientation)
    mk = lambda it, tol: _mk(ko, model, hq, q0, it, tol)
    A, B, Cc = mk(cap, 1e-5), mk(100, 1e-10), mk(100, 1e-10)
    out = dict(capped=[], iters_a=[], iters_b=[], dacc=[], ncon=[], drift=[], nefc=[])
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
            qa, qb = A.view("qacc"), B.view("qacc")
            out["capped"].append(int(A.s.newton_converged == 0 and A.s.nefc > 0))
            out["iters_a"].append(A.s.newton_iters_used)
            out["iters_b"].append(B.s.newton_iters_used)
            out["dacc"].append(np.abs(qa - qb).max() / (1.0 + np.abs(qb).max()))
            out["ncon"].append(A.s.ncon)
            out["nefc"].append(A.s.nefc)
            qc = Cc.view("qpos")
            out["drift"].append(np.abs(A.view("qpos") - qc).max() / max(1e-3, np.abs(qc).max()))
    return (shape, orientation, row), {k: np.array(v) for k, v in out.items()}


def _mk(ko, model, hq, q0, it, tol):
    s = ko.OracleSim(model, hq, solver_iterations=it)
    s.s.solver_tolerance = tol
    s.s.rays_enabled = 0
    s.env_reset(q0)
    return s


def run(cap=6):
    from kinovagrasping_amd import scenarios
    jobs = [("CubeS", o, r, cap) for o in POSES for r in (0, 700, 1400, 2100)]
    jobs += [(sh, "normal", r, cap) for sh in scenarios.SHAPES for r in (100, 1500)]
    with ProcessPoolExecutor(8) as ex:
        return list(ex.map(scenario, jobs))


def summarize(res, cap=6):
    lines = []
    cat = lambda k, sel=lambda key: True: np.concatenate([r[k] for key, r in res if sel(key)])
    lines.append(f"Newton cap {cap} (step tolerance 1e-5: the product kernels' and the parity oracle's setting) against cap 100 / tolerance 1e-10 on the SAME problems;")
    lines.append(f"{len(res)} closing-grasp + lift trajectories x 330 substeps = {len(cat('capped'))} substeps (fp64 oracle)")
    lines.append("")
    lines.append(f"{'scenario set':38s} {'substeps':>8s} {'exit at cap':>12s} {'it(cap) mean/max':>17s} {'it(100) mean/p99/max':>21s} {'rel |dqacc| p50 / p99 / max':>30s} {'qpos drift @200 / @330 (max)':>30s}")
    sets = [("CubeS normal (4 starts)", lambda k: k[0] == "CubeS" and k[1] == "normal" and k[2] not in (100, 1500)), ("CubeS rotated (4 starts)", lambda k: k[1] == "rotated"),
            ("CubeS top (4 starts)", lambda k: k[1] == "top"), ("14 shapes, normal (2 starts each)", lambda k: k[1] == "normal" and k[2] in (100, 1500)),
            ("all", lambda k: True)]
    for name, sel in sets:
        cp, ia, ib, da = cat("capped", sel), cat("iters_a", sel), cat("iters_b", sel), cat("dacc", sel)
        d200 = max(r["drift"][199] for key, r in res if sel(key))
        d330 = max(r["drift"][329] for key, r in res if sel(key))
        lines.append(f"{name:38s} {len(cp):8d} {cp.mean() * 100:11.2f}% {ia.mean():9.2f} /{ia.max():3d}    {ib.mean():8.2f} /{np.percentile(ib, 99):4.0f} /{ib.max():4d}   "
                     f"{np.median(da):9.1e} /{np.percentile(da, 99):9.1e} /{da.max():9.1e}   {d200:12.1e} /{d330:10.1e}")
    lines.append("")
    worst = sorted(res, key=lambda kr: -kr[1]["dacc"].max())[:5]
    lines.append("largest single-substep deviations:")
    for key, r in worst:
        i = int(r["dacc"].argmax())
        lines.append(f"  {key[0]:10s} {key[1]:8s} row {key[2]:5d}: substep {i:3d}, rel |dqacc| {r['dacc'][i]:.2e}, {r['ncon'][i]} contacts / {r['nefc'][i]} rows, "
                     f"{r['iters_b'][i]} iterations to converge, trajectory drift @330 {r['drift'][329]:.1e}")
    return "\n".join(lines)


if __name__ == "__main__":
    from kinovagrasping_amd.sim import SOLVER_ITERATIONS
    for cap in sorted({6, 8, 10, SOLVER_ITERATIONS}):
        print(summarize(run(cap), cap))
        print()
