#!/usr/bin/env python3
"""Row-22 study: single-parameter hypotheses about the finger-box contact of ONE substep (depth, normal yaw, normal tilt, position,
friction, R scale): fit the parameter + the 4 commands to row 22 (1 unknown vs 4 predicted numbers = consistency test)."""
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
def var(name, n=None):
    return (ctypes.c_double * n).in_dll(L, name) if n else ctypes.c_double.in_dll(L, name)
V = {"ddist": var("ko_dbg_ddist"), "rotz": var("ko_dbg_rotz"), "tilt": var("ko_dbg_tilt"), "mu": var("ko_dbg_mu"), "Rscale": var("ko_dbg_Rscale")}
dpos = var("ko_dbg_dpos", 3)
DEF = {"ddist": 0.0, "rotz": 0.0, "tilt": 0.0, "mu": 0.0, "Rscale": 1.0}

def sim_row(r, u, T, params):
    s.set_state(*states[r - 1])
    for k in range(4):
        if k in T:
            for n, v in params.items():
                if n.startswith("dpos"): dpos[int(n[4])] = v
                else: V[n].value = v
        s.step(old_env.ctrl_of(u))
        for n in DEF: V[n].value = DEF[n]
        for i in range(3): dpos[i] = 0.0
    return old_env.oracle_row(s)

COLS = [21, 22, 23, 28, 24, 25, 26, 27]
if __name__ == "__main__":
    r = 22
    for T in ([1], [2], [1, 2]):
        for names, x0, sc in ((["ddist"], [0.0], [1e-5]), (["rotz"], [0.0], [1e-2]), (["tilt"], [0.0], [1e-2]), (["mu"], [1.0], [0.1]), (["Rscale"], [1.0], [0.1]),
                              (["dpos0"], [0.0], [1e-3]), (["dpos1"], [0.0], [1e-3]), (["dpos2"], [0.0], [1e-3]), (["ddist", "rotz"], [0.0, 0.0], [1e-5, 1e-2]),
                              (["ddist", "Rscale"], [0.0, 1.0], [1e-5, 0.1])):
            def resid(x):
                row = sim_row(r, x[:4], T, dict(zip(names, x[4:])))
                return (row - pf2[r])[COLS]
            xx0 = np.concatenate([us[r], x0])
            try:
                sol = least_squares(resid, xx0, x_scale=np.array([0.01] * 4 + sc), diff_step=1e-6, xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=60)
                print(f"T={T} {names}: {sol.x[4:]}  resid box {np.round(sol.fun[:3],8)} dist {sol.fun[3]:.2e} | max {np.abs(sol.fun).max():.2e}")
            except Exception as e:
                print(T, names, "failed", e)
