#!/usr/bin/env python3
"""Round-5 study (CPU, scratch oracle of make_debug_oracle.py): are rows 46-62 of the recorded MuJoCo trajectory what the oracle gives for SOME
choice of the ties of the box's support function?  MuJoCo's analytic box support decides a direction component of 1e-17 by its sign
(`dir[i] > 0 ? 1 : -1`): rounding residue.  Here the sign pattern of the box's support skew (8 codes: which corner wins a tie) becomes a free
choice per (substep, finger geom touching the box) and is searched per row by coordinate descent on the row's miss of the recording
(commands refitted at the end of every sweep).  A row that drops from 1e-4 to 1e-8 under some pattern is a row the physics explains;
36 bits of choice cannot fit 44 numbers to 4 more digits by accident.
usage: python tools/r05/tie_search.py [first_row] [last_row] > profiles/r05_tie_search.txt"""
import sys, ctypes, itertools
import numpy as np
from _replay import load
sys.path.insert(0, "/tmp/dbg")
from tests import old_env
from oracle import ko_py
assert "/tmp/dbg" in ko_py.__file__, "run tools/r05/make_debug_oracle.py first"
pf2, rows0, us0, states0 = load()
L = ko_py.lib()
code = (ctypes.c_int * 16).in_dll(L, "ko_dbg_tie_code")
s = old_env.new_oracle_sim()
PRED = old_env.PREDICTED_COLS
PAIRS = (3, 5, 7)                       # f1_dist, f2_dist, f3_dist: the geoms on the box in these rows


def run_row(st, u, pat):
    """pat [4][3]: tie code per substep and pair"""
    s.set_state(*st)
    for k in range(4):
        for j, g in enumerate(PAIRS):
            code[g] = int(pat[k][j])
        s.step(old_env.ctrl_of(u))
    for g in PAIRS:
        code[g] = 0
    return old_env.oracle_row(s)


def miss(row, r):
    return np.abs(row[24:28] - pf2[r, 24:28]).max() + np.abs(row[PRED] - pf2[r, PRED]).max()


def refit(st, u, r, pat):
    """the 4 x 4 command problem under the pattern (old_env.recover_commands with the pattern applied)"""
    u = u.copy()
    for _ in range(10):
        row = run_row(st, u, pat); res = row[24:28] - pf2[r, 24:28]
        if np.abs(res).max() < 1e-12: break
        J = np.zeros((4, 4))
        for k in range(4):
            h = 1e-6 if u[k] < old_env.U_HI[k] - 1e-6 else -1e-6
            u2 = u.copy(); u2[k] += h
            J[:, k] = (run_row(st, u2, pat)[24:28] - row[24:28]) / h
        free = np.ones(4, bool); du = np.zeros(4)
        for _ in range(4):
            du = np.zeros(4); du[free] = np.linalg.lstsq(J[:, free], -res, rcond=None)[0]
            out = free & ((u + du < old_env.U_LO - 1e-15) | (u + du > old_env.U_HI + 1e-15))
            if not out.any(): break
            free &= ~out
        un = np.clip(u + du, old_env.U_LO, old_env.U_HI)
        if np.abs(un - u).max() < 1e-14: break
        u = un
    return u, run_row(st, u, pat)


r0 = int(sys.argv[1]) if len(sys.argv) > 1 else 46
r1 = int(sys.argv[2]) if len(sys.argv) > 2 else 62
st = states0[r0 - 1]
u = us0[r0 - 1].copy()
rng = np.random.default_rng(0)
found = {}
print("row   default-pattern miss   best miss   pattern (substep x [f1d f2d f3d])   commands")
for r in range(r0, r1 + 1):
    pat = np.zeros((4, 3), dtype=int)
    u_d, row_d = refit(st, u, r, pat)
    m_default = miss(row_d, r)
    best = (m_default, pat.copy(), u_d, row_d)
    for restart in range(int(sys.argv[3]) if len(sys.argv) > 3 else 6):
        pat = best[1].copy() if restart == 0 else rng.integers(0, 8, (4, 3))
        u_c, row_c = refit(st, u_d, r, pat)
        m_c = miss(row_c, r)
        improved = True
        while improved and m_c > 1e-8:
            improved = False
            for k, j in itertools.product(range(4), range(3)):
                keep = pat[k][j]
                for c in range(8):
                    if c == keep: continue
                    pat[k][j] = c
                    m = miss(run_row(st, u_c, pat), r)
                    if m < m_c * 0.999:
                        m_c, keep, improved = m, c, True
                pat[k][j] = keep
            u_c, row_c = refit(st, u_c, r, pat)
            m_c = miss(row_c, r)
        if m_c < best[0]:
            best = (m_c, pat.copy(), u_c, row_c)
        if best[0] < 1e-8:
            break
    m_b, pat_b, u_b, row_b = best
    print(f"{r:3d}   {m_default:10.2e}        {m_b:10.2e}   {' '.join(''.join(str(c) for c in p) for p in pat_b)}   {np.round(u_b, 4)}", flush=True)
    found[r] = {"miss": float(m_b), "pattern": pat_b.tolist(), "u": u_b.tolist()}
    import json; json.dump(found, open("/tmp/tie_patterns.json", "w"))
    run_row(st, u_b, pat_b)
    # re-run to leave the sim at the row's end state under the best pattern
    s.set_state(*st)
    for k in range(4):
        for j, g in enumerate(PAIRS): code[g] = int(pat_b[k][j])
        s.step(old_env.ctrl_of(u_b))
    for g in PAIRS: code[g] = 0
    st = old_env.oracle_state(s); u = u_b
