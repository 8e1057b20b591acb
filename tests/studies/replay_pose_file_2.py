#!/usr/bin/env python3
"""Study (CPU, test infrastructure - the oracle is the checker): the reference's recorded MuJoCo 1.50 contact trajectory
(Old Code/Pose_file_2.csv, fixture tests/golden/mujoco_recorded.npz: pose_file_2) replayed by the fp64 oracle on the emulated old
model (tests/old_env.py).  Prints, per row, the recovered commands and the error of every column group against the recording.
usage: python -m tests.studies.replay_pose_file_2 > profiles/r04_pose_file_2_replay.txt"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests import old_env  # noqa: E402

GROUPS = [("finger-link centres (18)", slice(0, 18)), ("palm centre (3)", slice(18, 21)), ("object xyz (3)", slice(21, 24)),
          ("actuated joints (4)", slice(24, 28)), ("distal joints (3)", slice(28, 31)), ("site-object dist (13)", slice(34, 47)), ("dot^20", slice(47, 48))]


def main():
    pf2 = np.load(ROOT / "tests" / "golden" / "mujoco_recorded.npz")["pose_file_2"]
    rows, us, _ = old_env.replay_recording(pf2)
    err = np.abs(rows - pf2)
    print("Old Code/Pose_file_2.csv (real MuJoCo 1.50, 63 rows x 48 columns) replayed by the fp64 oracle; errors = |oracle - recording| per column group")
    print("commands [wrist, f1, f2, f3] recovered per row from the four actuated joint angles (tests/old_env.py); all other columns are predictions\n")
    print("row   wrist    f1      f2      f3    | " + " | ".join(f"{g[0]:>24s}" for g in GROUPS) + " | object xyz recorded")
    for r in range(len(pf2)):
        print(f"{r:3d}  {us[r][0]:.4f}  {us[r][1]:.4f}  {us[r][2]:.4f}  {us[r][3]:.4f} | " + " | ".join(f"{err[r, g[1]].max():24.2e}" for g in GROUPS) +
              " | " + " ".join(f"{x:8.5f}" for x in pf2[r, 21:24]))
    beyond = np.nonzero(err[:, :47].max(1) > 1e-6)[0]
    r0 = int(beyond[0])
    print(f"\nrows 0-{r0 - 1}: max error over all 48 columns {err[:r0].max():.2e} (columns 0-46: {err[:r0, :47].max():.2e})")
    print(f"first row beyond 1e-6: row {r0}, column {int(err[r0, :47].argmax())} ({err[r0, :47].max():.2e}); object xyz error there {err[r0, 21:24]}")
    print(f"rows {r0}-62: object xyz max {err[r0:, 21:24].max(0)}, finger-link centres {err[r0:, :18].max():.2e}, distal joints {err[r0:, 28:31].max():.2e}, "
          f"actuated joints median {np.median(err[r0:, 24:28].max(1)):.1e} max {err[r0:, 24:28].max():.2e}")
    print(f"final row: object height recorded {pf2[62, 23]:.5f}, replayed {rows[62, 23]:.5f}")


if __name__ == "__main__":
    main()
