#!/usr/bin/env python3
"""Row-22 study: KKT residual of the oracle's Newton solution in every substep of rows 19-24."""
import sys, pickle
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests import old_env
rows, us, states = pickle.load(open("/tmp/replay_cache.pkl", "rb"))
s = old_env.new_oracle_sim()
for r in range(19, 25):
    s.set_state(*states[r - 1])
    for k in range(4):
        s.view("ctrl")[:] = old_env.ctrl_of(us[r])
        s.forward()
        n = s.s.nefc
        M = s.view("M").reshape(15, 15); J = s.view("efc_J").reshape(-1, 15)[:n]; R = s.view("efc_R")[:n]; aref = s.view("efc_aref")[:n]
        ty = s.view("efc_type")[:n]
        a = s.view("qacc").copy(); a_s = s.view("qacc_smooth")
        jar = J @ a - aref
        act = (ty == 0) | (jar < 0)
        f = np.where(act, -jar / R, 0.0)
        g = M @ (a - a_s) - J.T @ f
        print(r, k, "nefc", n, "iters", s.s.newton_iters_used, "conv", s.s.newton_converged, "|g|", np.abs(g[2:]).max(), "min|jar| of rows", np.abs(jar).min() if n else None)
        s.step(old_env.ctrl_of(us[r]))
