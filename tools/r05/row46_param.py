#!/usr/bin/env python3
"""Row-46 study: single-parameter hypotheses on the finger-1 / box contact of ONE substep among (45,3), (46,0), (46,1), (46,2)."""
import sys, pickle, ctypes
from pathlib import Path
import numpy as np
from scipy.optimize import least_squares
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, "/tmp/dbg")
from tests import old_env
from oracle import ko_py
pf2 = np.load(ROOT / "tests/golden/mujoco_recorded.npz")["pose_file_2"]
rows, us, states = pickle.load(open("/tmp/replay_ms.pkl", "rb"))
L = ko_py.lib()
ctypes.c_int.in_dll(L, "ko_dbg_obj_first").value = 1
import os
ctypes.c_int.in_dll(L, "ko_dbg_g1").value = int(os.environ.get("G1", "3"))
s = old_env.new_oracle_sim()
def var(name, n=None):
    return (ctypes.c_double * n).in_dll(L, name) if n else ctypes.c_double.in_dll(L, name)
V = {"frot": var("ko_dbg_frot"), "ddist": var("ko_dbg_ddist"), "rotz": var("ko_dbg_rotz"), "tilt": var("ko_dbg_tilt"), "mu": var("ko_dbg_mu"), "Rscale": var("ko_dbg_Rscale")}
dpos = var("ko_dbg_dpos", 3)
DEF = {"frot": 0.0, "ddist": 0.0, "rotz": 0.0, "tilt": 0.0, "mu": 0.0, "Rscale": 1.0}
R0 = int(sys.argv[1]) if len(sys.argv) > 1 else 45      # first row simulated
NR = int(sys.argv[2]) if len(sys.argv) > 2 else 2

def sim(ulist, K, params):
    s.set_state(*states[R0 - 1])
    out = []
    g = 0
    for i, u in enumerate(ulist):
        for k in range(4):
            if g == K:
                for n, v in params.items():
                    if n.startswith("dpos"): dpos[int(n[4])] = v
                    else: V[n].value = v
            s.step(old_env.ctrl_of(np.clip(u, old_env.U_LO, old_env.U_HI)))
            for n in DEF: V[n].value = DEF[n]
            for j in range(3): dpos[j] = 0.0
            g += 1
        out.append(old_env.oracle_row(s))
    return np.array(out)

COLS = [21, 22, 23, 28, 29, 30, 24, 25, 26, 27]
if __name__ == "__main__":
    base = sim([us[R0 + i] for i in range(NR)], -1, {})
    print("baseline resid rows", [f"{np.abs((base[i]-pf2[R0+i])[COLS]).max():.2e}" for i in range(NR)])
    for K in range(2, 4 * NR - 1):
        for names, x0, sc in ((["frot"], [0.3], [0.1]), (["frot"], [-0.3], [0.1]), (["ddist"], [0.0], [1e-5]), (["rotz"], [0.0], [1e-2]), (["tilt"], [0.0], [1e-2]), (["dpos0"], [0.0], [1e-3]), (["dpos1"], [0.0], [1e-3]), (["dpos2"], [0.0], [1e-3]),
                              (["rotz", "ddist"], [0.0, 0.0], [1e-2, 1e-5]), (["rotz", "tilt", "ddist"], [0.0, 0.0, 0.0], [1e-2, 1e-2, 1e-5])):
            def resid(x):
                ul = [us[R0 + i].copy() for i in range(NR)]
                ul[-1] = x[:4]
                rr = sim(ul, K, dict(zip(names, x[4:])))
                return np.concatenate([(rr[i] - pf2[R0 + i])[COLS] for i in range(NR)])
            xx0 = np.concatenate([us[R0 + NR - 1], x0])
            try:
                sol = least_squares(resid, xx0, x_scale=np.array([0.01] * 4 + sc), diff_step=1e-6, xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=80)
                print(f"K={K} (row {R0 + K // 4} substep {K % 4}) {names}: {sol.x[4:]} | max resid {np.abs(sol.fun).max():.2e}  u {np.round(sol.x[:4], 4)}")
            except Exception as e:
                print(K, names, "failed", e)
