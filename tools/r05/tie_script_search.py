#!/usr/bin/env python3
"""Round-5 study (CPU, scratch oracle of make_debug_oracle.py), second form of tools/r05/tie_search.py: EVERY near-zero component of a box support
direction (|component| < 1e-9 of the direction: a tie between box corners that MuJoCo's `dir[i] > 0 ? 1 : -1` decides by rounding residue) takes its
sign from a script - one entry per tie event, in the order the events occur within a substep.  Per row the four substeps' scripts are searched by
coordinate descent (flip one entry, keep it if the row's miss of the recording drops), commands refitted after every sweep, random restarts.
usage: python tools/r05/tie_script_search.py [first_row] [last_row] > profiles/r05_tie_script_search.txt"""
import sys, ctypes, json
import numpy as np
from _replay import load
sys.path.insert(0, "/tmp/dbg")
from tests import old_env
from oracle import ko_py
assert "/tmp/dbg" in ko_py.__file__, "run tools/r05/make_debug_oracle.py first"
pf2, rows0, us0, states0 = load()
L = ko_py.lib()
script_c = (ctypes.c_int * 256).in_dll(L, "ko_dbg_tie_script")
tie_n = ctypes.c_int.in_dll(L, "ko_dbg_tie_n")
tie_mode = ctypes.c_int.in_dll(L, "ko_dbg_tie_mode")
s = old_env.new_oracle_sim()
PRED = old_env.PREDICTED_COLS
NS = 64


def run_row(st, u, scr, used=None):
    """scr [4][NS] 0/1; used: optional list receiving the number of tie events per substep"""
    s.set_state(*st)
    tie_mode.value = 1
    for k in range(4):
        for i in range(NS):
            script_c[i] = int(scr[k][i])
        tie_n.value = 0
        s.step(old_env.ctrl_of(u))
        if used is not None:
            used.append(tie_n.value)
    tie_mode.value = 0
    return old_env.oracle_row(s)


def miss(row, r):
    return np.abs(row[24:28] - pf2[r, 24:28]).max() + np.abs(row[PRED] - pf2[r, PRED]).max()


def refit(st, u, r, scr):
    u = u.copy()
    for _ in range(10):
        row = run_row(st, u, scr); res = row[24:28] - pf2[r, 24:28]
        if np.abs(res).max() < 1e-12: break
        J = np.zeros((4, 4))
        for k in range(4):
            h = 1e-6 if u[k] < old_env.U_HI[k] - 1e-6 else -1e-6
            u2 = u.copy(); u2[k] += h
            J[:, k] = (run_row(st, u2, scr)[24:28] - row[24:28]) / h
        free = np.ones(4, bool); du = np.zeros(4)
        for _ in range(4):
            du = np.zeros(4); du[free] = np.linalg.lstsq(J[:, free], -res, rcond=None)[0]
            out = free & ((u + du < old_env.U_LO - 1e-15) | (u + du > old_env.U_HI + 1e-15))
            if not out.any(): break
            free &= ~out
        un = np.clip(u + du, old_env.U_LO, old_env.U_HI)
        if np.abs(un - u).max() < 1e-14: break
        u = un
    return u, run_row(st, u, scr)


r0 = int(sys.argv[1]) if len(sys.argv) > 1 else 46
r1 = int(sys.argv[2]) if len(sys.argv) > 2 else 62
st = states0[r0 - 1]
u = us0[r0 - 1].copy()
rng = np.random.default_rng(1)
found = {}
print("row   miss with script 0   best miss   tie events per substep   entries set   commands")
for r in range(r0, r1 + 1):
    scr = np.zeros((4, NS), dtype=int)
    u_d, row_d = refit(st, u, r, scr)
    best = (miss(row_d, r), scr.copy(), u_d)
    m0 = best[0]
    for restart in range(10):
        scr = best[1].copy() if restart == 0 else rng.integers(0, 2, (4, NS))
        u_c, row_c = refit(st, u_d, r, scr)
        m_c = miss(row_c, r)
        improved = True
        while improved and m_c > 1e-8:
            improved = False
            used = []
            run_row(st, u_c, scr, used)
            for k in range(4):
                for i in range(min(used[k], NS)):
                    scr[k][i] ^= 1
                    m = miss(run_row(st, u_c, scr), r)
                    if m < m_c * 0.999:
                        m_c, improved = m, True
                    else:
                        scr[k][i] ^= 1
            u_c, row_c = refit(st, u_c, r, scr)
            m_c = miss(row_c, r)
        if m_c < best[0]:
            best = (m_c, scr.copy(), u_c)
        if best[0] < 1e-8:
            break
    m_b, scr_b, u_b = best
    used = []
    run_row(st, u_b, scr_b, used)
    st = old_env.oracle_state(s); u = u_b
    found[r] = {"miss": m_b, "used": used, "script": [scr_b[k][:used[k]].tolist() for k in range(4)], "u": u_b.tolist()}
    print(f"{r:3d}   {m0:10.2e}          {m_b:10.2e}   {used}   {[int(scr_b[k][:used[k]].sum()) for k in range(4)]}   {np.round(u_b, 4)}", flush=True)
json.dump(found, open("/tmp/tie_scripts.json", "w"))
