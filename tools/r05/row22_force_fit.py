#!/usr/bin/env python3
"""Row-22 study: replace the finger-box contact of ONE substep of row 22 by a prescribed force (contact frame of the oracle) and
fit it (+ the 4 commands) to the recording's row 22.  3 force unknowns vs 4 predicted numbers (box xyz, f1 distal): a consistency test."""
import sys, pickle, ctypes
from pathlib import Path
import numpy as np
from scipy.optimize import least_squares
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, "/tmp/dbg")
from tests import old_env
from oracle import ko_py
pf2 = np.load(ROOT / "tests/golden/mujoco_recorded.npz")["pose_file_2"]
rows, us, states = pickle.load(open("/tmp/replay_cache.pkl", "rb"))
s = old_env.new_oracle_sim()
L = ko_py.lib()
qfrc = (ctypes.c_double * 15).in_dll(L, "ko_dbg_qfrc")
dg1 = ctypes.c_int.in_dll(L, "ko_dbg_drop_g1"); dg2 = ctypes.c_int.in_dll(L, "ko_dbg_drop_g2")

def sim_row22(u, T, F):
    s.set_state(*states[21])
    for k in range(4):
        s.view("ctrl")[:] = old_env.ctrl_of(u)
        if k in T:
            s.forward()
            n = s.s.nefc
            ty = s.view("efc_type")[:n]
            J = s.view("efc_J").reshape(-1, 15)[:n]
            cons = s.contacts()
            idx = [i for i, c in enumerate(cons) if c["geom1"] == 3]
            assert len(idx) == 1
            row0 = int((ty != 2).sum()) + 4 * idx[0]
            Jn = 0.5 * (J[row0] + J[row0 + 1]); Jt1 = 0.5 * (J[row0] - J[row0 + 1]); Jt2 = 0.5 * (J[row0 + 2] - J[row0 + 3])
            Fk = F[3 * T.index(k):3 * T.index(k) + 3]
            q = Jn * Fk[0] + Jt1 * Fk[1] + Jt2 * Fk[2]
            for i in range(15): qfrc[i] = q[i]
            dg1.value, dg2.value = 3, 8
        s.step(old_env.ctrl_of(u))
        for i in range(15): qfrc[i] = 0.0
        dg1.value = dg2.value = -1
    return old_env.oracle_row(s)

COLS = [21, 22, 23, 24, 25, 26, 27, 28, 29, 30]
def resid(x, T):
    row = sim_row22(x[:4], T, x[4:])
    return (row - pf2[22])[COLS]

if __name__ == "__main__":
    F1 = [1.2018, 0.1297, 1.0721]; F2 = [0.2566, 0.0, -0.2566]
    for T, F0 in (([1], F1), ([2], F2), ([1, 2], F1 + F2)):
        x0 = np.concatenate([us[22], F0])
        r0 = resid(x0, T)
        print("T", T, "baseline resid", np.round(r0, 9))
        sol = least_squares(resid, x0, args=(T,), x_scale=np.array([0.01] * 4 + [0.1] * (3 * len(T))), diff_step=1e-7, xtol=1e-15, ftol=1e-15, gtol=1e-15)
        print("   fit: u", sol.x[:4], "F", sol.x[4:], "\n   resid", np.round(sol.fun, 10), "max", np.abs(sol.fun).max())
