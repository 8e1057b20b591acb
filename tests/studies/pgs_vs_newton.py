#!/usr/bin/env python3
"""Convergence of PGS (north_star's prescription) vs Newton (MuJoCo's default for this XML) on the
oracle: max |qacc - qacc_converged| at states of a CubeS grasp run.  Output: profiles/r01_pgs_vs_newton.txt"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import ko_py as ko
from kinovagrasping_amd import scenarios

blob = scenarios.model_blob("CubeS")
m = ko.OracleModel(blob)
hq = scenarios.hand_quat_for("normal")
s = ko.OracleSim(m, hq, solver_iterations=30)
q0 = np.zeros(16); q0[9:12] = [0.052897, 0.000732, 0.0654]; q0[12] = 1
s.env_reset(q0)
ctrl = np.zeros(9); ctrl[5] = 0.2932; ctrl[6:9] = 0.3
states = []
for i in range(200):
    s.step(ctrl)
    if i in (7, 30, 100, 199):
        states.append((i, s.view("qpos").copy(), s.view("qvel").copy(), s.view("qacc_warmstart").copy()))
lines = ["state(substep) rows  solver  iterations  max|qacc - converged|"]
for i, qp, qv, qw in states:
    s.s.solver = 0; s.s.solver_iterations = 60
    s.set_state(qp, qv, qw); s.view("ctrl")[:] = ctrl; s.forward(); ref = s.view("qacc").copy()
    for solver, name, its in ((0, "newton", (1, 2, 3, 4, 6)), (1, "pgs", (10, 50, 100, 500, 1000, 2000))):
        for it in its:
            s.s.solver = solver; s.s.solver_iterations = it
            s.set_state(qp, qv, qw); s.forward()
            lines.append(f"{i:5d} {s.s.nefc:4d}  {name:6s} {it:6d}  {np.abs(s.view('qacc') - ref).max():.3e}")
out = ROOT / "profiles" / "r01_pgs_vs_newton.txt"
out.write_text("\n".join(lines) + "\n")
print("\n".join(lines))
