#!/usr/bin/env python3
"""Row-22 study: MPR internals (debug copy of the oracle under /tmp/dbg) for rows r0..r1."""
import sys, pickle, ctypes
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, "/tmp/dbg")
from tests import old_env
from oracle import ko_py
assert "/tmp/dbg" in ko_py.__file__
rows, us, states = pickle.load(open("/tmp/replay_cache.pkl", "rb"))
s = old_env.new_oracle_sim()
tr = ctypes.c_int.in_dll(ko_py.lib(), "ko_trace")
for r in range(int(sys.argv[1]), int(sys.argv[2])):
    s.set_state(*states[r - 1])
    for k in range(4):
        print(f"row {r} substep {k}", file=sys.stderr)
        tr.value = 1
        s.step(old_env.ctrl_of(us[r]))
        tr.value = 0
