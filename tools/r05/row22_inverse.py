#!/usr/bin/env python3
"""Row-22 study, the inverse problem (VERDICT r4 next #1a): from the matched row-21 state, which extra impulse (object 6 dof +
finger-1 distal joint), applied after ONE substep K, reproduces the recording's rows 22 (and 23)?  CPU, oracle = checker."""
import sys, pickle
from pathlib import Path
import numpy as np
from scipy.optimize import least_squares
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests import old_env
pf2 = np.load(ROOT / "tests/golden/mujoco_recorded.npz")["pose_file_2"]
rows, us, states = pickle.load(open("/tmp/replay_cache.pkl", "rb"))
s = old_env.new_oracle_sim()
COLS = list(range(0, 24)) + list(range(24, 31)) + list(range(34, 47))
W = np.ones(48); 

def run_rows(K, kick, nrows=2, refit=True):
    """K = global substep index counted from row 21 substep 0 (0..11). kick = dv for qvel[9:15] + qvel[4]"""
    st = states[20]
    out = []
    u_prev = us[21].copy()
    sub = 0
    for r in range(21, 22 + nrows):
        u = us[r].copy()
        def sim_row(st, u, sub0):
            s.set_state(*st)
            for k in range(4):
                s.step(old_env.ctrl_of(u))
                if sub0 + k == K:
                    s.view("qvel")[9:15] += kick[:6]
                    s.view("qvel")[4] += kick[6]
            return old_env.oracle_row(s)
        if refit and r >= 22:
            for _ in range(8):
                row = sim_row(st, u, sub)
                res = row[24:28] - pf2[r, 24:28]
                if np.abs(res).max() < 1e-12: break
                J = np.zeros(4)
                for k in range(4):
                    h = 1e-5 if u[k] < old_env.U_HI[k] - 1e-5 else -1e-5
                    u2 = u.copy(); u2[k] += h
                    J[k] = (sim_row(st, u2, sub)[24 + k] - row[24 + k]) / h
                ok = np.abs(J) > 1e-9
                u = np.clip(u - np.where(ok, res / np.where(ok, J, 1.0), 0.0), old_env.U_LO, old_env.U_HI)
        row = sim_row(st, u, sub)
        st = old_env.oracle_state(s)
        sub += 4
        out.append(row)
    return np.array(out)

def resid(kick, K, nrows):
    R = run_rows(K, kick, nrows)
    return ((R[1:] - pf2[22:22 + nrows])[:, COLS]).ravel()

if __name__ == "__main__":
    nrows = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    base = resid(np.zeros(7), -1, nrows)
    print("baseline residual max", np.abs(base).max())
    for K in range(3, 7):
        sol = least_squares(resid, np.zeros(7), args=(K, nrows), x_scale=1e-3, diff_step=1e-6, xtol=1e-15, ftol=1e-15, gtol=1e-15)
        print(f"K={K} (row {21 + K // 4} substep {K % 4}): |res|max {np.abs(sol.fun).max():.3e}  kick lin {sol.x[:3]} ang {sol.x[3:6]} dist {sol.x[6]:.3e}")
