#!/usr/bin/env python3
"""Row-22 study: linear response of row 22's predicted numbers to the finger-box contact force of substeps 1 and 2 (forces
prescribed in the oracle's contact frame, contact constraint removed), commands refit by the same linearisation."""
import sys, pickle, ctypes
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, "/tmp/dbg")
from tests import old_env
from oracle import ko_py
from tools.r05.row22_force_fit import sim_row22, pf2, us
np.set_printoptions(linewidth=200, precision=4, suppress=False)
F1 = [1.2018, 0.1297, 1.0721]; F2 = [0.2566, 0.0, -0.2566]
# exact forces from the oracle
x0 = np.concatenate([us[22], F1, F2])
OBS = [21, 22, 23, 28, 24, 25, 26, 27]     # box xyz, f1 distal | 4 actuated
def f(x): return sim_row22(x[:4], [1, 2], x[4:])[OBS]
y0 = f(x0)
tgt = pf2[22][OBS]
print("baseline resid", y0 - tgt)
J = np.zeros((8, 10))
for i in range(10):
    h = 1e-6
    xp = x0.copy(); xp[i] += h
    xm = x0.copy(); xm[i] -= h
    J[:, i] = (f(xp) - f(xm)) / (2 * h)
print("J (rows: box x y z, f1dist, wrist, f1p, f2p, f3p; cols: u0..u3, F1 n t1 t2, F2 n t1 t2)\n", J)
r = tgt - y0
# eliminate commands: actuated rows must stay matched
A_u = J[4:, :4]; A_f = J[4:, 4:]
# du = -A_u^-1 A_f dF ; effect on predicted rows:
S = J[:4, 4:] - J[:4, :4] @ np.linalg.solve(A_u, A_f)
print("reduced sensitivity S (4 x 6): d[box xyz, f1dist]/d[F1 n t1 t2, F2 n t1 t2]\n", S)
for name, cols in (("substep 1 only", [0, 1, 2]), ("substep 2 only", [3, 4, 5]), ("both", [0, 1, 2, 3, 4, 5])):
    dF, res, rk, sv = np.linalg.lstsq(S[:, cols], r[:4], rcond=None)
    print(name, "dF", dF, "remaining", S[:, cols] @ dF - r[:4])
