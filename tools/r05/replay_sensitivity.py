#!/usr/bin/env python3
"""Round-5 study (CPU, oracle): is the replay of the recorded MuJoCo trajectory sensitive to its state?  Random perturbations of qpos / qvel after
row r0, commands held: change of the object's position over the next three rows.  Result: linear, gain ~2, in every phase -
push (20, 30), grasp (38), lift (43-45, where the replay leaves the recording at row 46): no chaos anywhere.
usage: python tools/r05/replay_sensitivity.py > profiles/r05_replay_sensitivity.txt"""
import numpy as np
from _replay import load
from tests import old_env
pf2, rows, us, states = load()
s = old_env.new_oracle_sim()
rng = np.random.default_rng(0)
print("perturbation of qpos[0:12] / qvel after row r0 (4 random draws, commands held) -> max change of the object's xyz in rows r0+1 .. r0+3")
for r0 in (20, 30, 38, 43, 44, 45):
    for eps in (1e-12, 1e-10, 1e-8):
        outs = []
        for trial in range(4):
            q, v, w = [x.copy() for x in states[r0]]
            q[:12] += eps * rng.standard_normal(12); v += eps * rng.standard_normal(15)
            st, res = (q, v, w), []
            for r in range(r0 + 1, min(r0 + 4, 63)):
                row = old_env._run_row(s, st, us[r]); st = old_env.oracle_state(s)
                res.append(np.abs(row[21:24] - rows[r][21:24]).max())
            outs.append(res)
        print(f"row {r0:2d}  eps {eps:.0e}:  " + "  ".join(f"{x:.2e}" for x in np.max(outs, 0)))
