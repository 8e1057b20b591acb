#!/usr/bin/env python3
"""Row-22 study (CPU, oracle): per-substep contact list of rows 19-24 of the replayed recording."""
import sys, pickle
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests import old_env

GN = ["ground", "palm", "f1p", "f1d", "f2p", "f2d", "f3p", "f3d", "obj"]
pf2 = np.load(ROOT / "tests/golden/mujoco_recorded.npz")["pose_file_2"]
cache = Path("/tmp/replay_cache.pkl")
if cache.exists():
    rows, us, states = pickle.load(open(cache, "rb"))
else:
    rows, us, states = old_env.replay_recording(pf2, n_rows=40)
    pickle.dump((rows, us, states), open(cache, "wb"))
s = old_env.new_oracle_sim()
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 19, int(sys.argv[2]) if len(sys.argv) > 2 else 25):
    s.set_state(*states[r - 1])
    print(f"--- row {r} u={us[r]} err obj {rows[r,21:24]-pf2[r,21:24]} dist {rows[r,28:31]-pf2[r,28:31]}")
    for k in range(4):
        s.step(old_env.ctrl_of(us[r]))
        cf = s.contact_forces()
        print(f"  substep {k}: ncon {s.s.ncon} newton it {s.s.newton_iters_used} objz {s.view('qpos')[11]:.6f} quat {s.view('qpos')[12:16]}")
        for i, c in enumerate(s.contacts()):
            print(f"     {GN[c['geom1']]:>6s}-{GN[c['geom2']]:<4s} dist {c['dist']:+.3e} n {np.round(c['frame'][:3],4)} pos {np.round(c['pos'],5)} f {np.round(cf[i],4)}")
