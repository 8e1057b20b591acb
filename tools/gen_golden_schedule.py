#!/usr/bin/env python3
"""Dev-only: known answers of the reference's object scheduling (Generate_Latin_Square, kinova_gripper_env.py:895-964)
and orientation selection (select_orienation, 1180-1222) -> tests/golden/schedule.npz.  Same stub import as
tests/golden/gen_golden_env.py; the env object is built with __new__ (no MuJoCo)."""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
REF = Path("/root/reference/gym-kinova-gripper")
sys.path.insert(0, str(REPO / "tests" / "golden"))
from gen_golden_env import install_stubs  # noqa: E402


def main():
    install_stubs()
    os.chdir(REF)
    sys.path.insert(0, str(REF))
    sys.path.insert(0, str(REF / "gym_kinova_gripper" / "envs"))
    import kinova_gripper_env as kge
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for tag, keys, m in (("a", ["CubeS", "CubeB", "CylinderS", "Vase1B"], 11), ("b", ["CubeS"], 5), ("c", ["Cone1S", "Cone2B", "Cube45S"], 24),
                             ("d", ["CubeS", "CubeB", "CylinderS", "CylinderB", "Cube45S", "Cube45B", "Cone1S"], 30)):
            env = kge.KinovaGripper_Env.__new__(kge.KinovaGripper_Env)
            env.obj_keys = []
            env.objects = {}
            env.all_objects = {k: k + ".xml" for k in keys}
            env.Generate_Latin_Square(m, os.path.join(td, tag + ".csv"), shape_keys=keys)
            out[f"ls_{tag}_keys"] = np.array(keys)
            out[f"ls_{tag}_n"] = np.array(m)
            out[f"ls_{tag}_out"] = np.array(env.obj_keys)
    env = kge.KinovaGripper_Env.__new__(kge.KinovaGripper_Env)
    shapes = ["CubeS", "RBowlB", "LemonS", "Vase2B"] * 40
    modes = ["random"] * 120 + ["normal"] * 40
    np.random.seed(123)
    sel = [env.select_orienation(s, h) for s, h in zip(shapes, modes)]
    out["or_shapes"], out["or_modes"], out["or_out"] = np.array(shapes), np.array(modes), np.array(sel)
    dst = REPO / "tests" / "golden" / "schedule.npz"
    np.savez_compressed(dst, **out)
    print("wrote", dst, {k: v.shape for k, v in out.items() if k.endswith("out")}, sorted(set(sel)))


if __name__ == "__main__":
    main()
