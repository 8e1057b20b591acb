#!/usr/bin/env python3
"""Round-5 study (CPU, scratch oracle of make_debug_oracle.py): the penetration queries of rows r0..r1 of the replay - per query the portal normal
against libccd's witness direction (cos < 1: the closest point of the final portal triangle is on its EDGE) and, with level 2, every support
direction in the box's frame (a component below 1e-9 = a tie between box corners that MuJoCo's analytic support decides by rounding).
usage: python tools/r05/mpr_ties.py 44 47 [2] 2> trace.txt"""
import sys, ctypes
from _replay import load
sys.path.insert(0, "/tmp/dbg")
from tests import old_env
from oracle import ko_py
assert "/tmp/dbg" in ko_py.__file__, "run tools/r05/make_debug_oracle.py first"
pf2, rows, us, states = load()
tr = ctypes.c_int.in_dll(ko_py.lib(), "ko_trace")
s = old_env.new_oracle_sim()
for r in range(int(sys.argv[1]), int(sys.argv[2])):
    s.set_state(*states[r - 1])
    for k in range(4):
        print(f"row {r} substep {k}", file=sys.stderr)
        tr.value = int(sys.argv[3]) if len(sys.argv) > 3 else 1
        s.step(old_env.ctrl_of(us[r]))
        tr.value = 0
