#!/usr/bin/env python3
"""Row study: continue the multi-start replay from a cached row with debug switches of the scratch oracle (/tmp/dbg)."""
import sys, pickle, time, ctypes
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, "/tmp/dbg")
from tests import old_env
from oracle import ko_py
pf2 = np.load(ROOT / "tests/golden/mujoco_recorded.npz")["pose_file_2"]
rows0, us0, states0 = pickle.load(open("/tmp/replay_ms.pkl", "rb"))
r_start, r_end = int(sys.argv[1]), int(sys.argv[2])
L = ko_py.lib()
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    try: ctypes.c_int.in_dll(L, k).value = int(v)
    except ValueError: ctypes.c_double.in_dll(L, k).value = float(v)
s = old_env.new_oracle_sim()
PRED = [c for c in range(47) if c not in (24, 25, 26, 27, 31, 32, 33)]
def run_row(st, u):
    s.set_state(*st)
    for _ in range(4): s.step(old_env.ctrl_of(u))
    return old_env.oracle_row(s)
def newton(st, u, tgt, iters=14):
    """full 4 x 4 Jacobian (the fingers couple through the grasped box), saturated commands held at their bound"""
    u = u.copy()
    for _ in range(iters):
        row = run_row(st, u); res = row[24:28] - tgt
        if np.abs(res).max() < 1e-12: break
        J = np.zeros((4, 4))
        for k in range(4):
            h = 1e-6 if u[k] < old_env.U_HI[k] - 1e-6 else -1e-6
            u2 = u.copy(); u2[k] += h
            J[:, k] = (run_row(st, u2)[24:28] - row[24:28]) / h
        free = np.ones(4, bool)
        for _ in range(4):
            du = np.zeros(4)
            try:
                du[free] = np.linalg.lstsq(J[:, free], -res, rcond=None)[0]
            except np.linalg.LinAlgError:
                break
            un = u + du
            viol = free & ((un < old_env.U_LO - 1e-15) | (un > old_env.U_HI + 1e-15))
            if not viol.any(): break
            free &= ~viol
        un = np.clip(u + du, old_env.U_LO, old_env.U_HI)
        if np.abs(un - u).max() < 1e-14: break
        u = un
    return u, run_row(st, u)
st = states0[r_start - 1]; u = us0[r_start - 1].copy()
for r in range(r_start, r_end):
    tgt = pf2[r, 24:28]
    cands = []
    u1, row1 = newton(st, u, tgt)
    cands.append((np.abs(row1[24:28] - tgt).max() + np.abs(row1[PRED] - pf2[r, PRED]).max(), u1, row1))
    if cands[0][0] > 1e-8:
        for k in range(4):
            for v in np.linspace(old_env.U_LO[k], old_env.U_HI[k], 17):
                u0 = u1.copy(); u0[k] = v
                u2, row2 = newton(st, u0, tgt, iters=8)
                cands.append((np.abs(row2[24:28] - tgt).max() + np.abs(row2[PRED] - pf2[r, PRED]).max(), u2, row2))
    e, ub, rowb = min(cands, key=lambda c: c[0])
    run_row(st, ub); st = old_env.oracle_state(s); u = ub
    print(f"row {r:2d} u {np.round(ub,4)} act {np.abs(rowb[24:28]-tgt).max():.1e} pred {np.abs(rowb[PRED]-pf2[r,PRED]).max():.2e} obj {np.abs(rowb[21:24]-pf2[r,21:24]).max():.2e} dist {np.abs(rowb[28:31]-pf2[r,28:31]).max():.2e}", flush=True)
