#!/usr/bin/env python3
"""Dev-only: generate tests/golden/reset_helpers.npz by running the REFERENCE's own reset helpers (kinova_gripper_env.py) on a fake `_sim`:

  _get_obj_size                          (:706-746)   for all 42 objects of the object table, geom_size arrays of the compiled models, with the env's
                                                      size letter as the object schedule leaves it ('m') and as the obj_params hook sets it
  determine_hand_location                (:1286-1307) for the three orientation classes x the three size letters, Tfw from the reference's own
                                                      _get_trans_mat_wrist_pose on the palm pose of that orientation
  randomize_initial_pos_data_collection  (:821-849)   seeded np.random, three orientation names, several objects
  sample_initial_object_hand_pos         (:1008-1054) seeded np.random on the reference's own coordinate files: no region, the four x-regions
                                                      (index slip included), "origin"; no_noise and with_noise files

The reference cannot travel to the GPU box, so only these input / output vectors are committed.  Reads /root/reference by absolute path; needs no
MuJoCo (the stubs of gen_golden_env.py).  usage: python tests/golden/gen_golden_reset.py"""
import os
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from gen_golden_env import REF, REPO, install_stubs  # noqa: E402


def main():
    install_stubs()
    os.chdir(REF)
    sys.path.insert(0, str(REF))
    sys.path.insert(0, str(REF / "gym_kinova_gripper" / "envs"))
    import kinova_gripper_env as kge
    from kinovagrasping_amd import model_compiler as mc, scenarios

    def new_env():
        return kge.KinovaGripper_Env.__new__(kge.KinovaGripper_Env)

    out = {}
    # ---- _get_obj_size over the object table
    keys = scenarios.SHAPES + scenarios.MEDIUM_SHAPES + scenarios.EXTRA_SHAPES + scenarios.MULTI_GEOM_SHAPES
    # the table itself comes from the reference's source (all_objects is filled in __init__: read its assignments)
    import re
    src = (REF / "gym_kinova_gripper" / "envs" / "kinova_gripper_env.py").read_text()
    table = dict(re.findall(r'self\.all_objects\["([A-Za-z0-9]+)"\]\s*=\s*"([^"]+)"', src))
    assert sorted(table) == sorted(keys)
    sizes_sched, sizes_hook, names = [], [], []
    for k in keys:
        M = mc.read_blob(scenarios.model_blob(k))
        env = new_env()
        env._sim = SimpleNamespace(model=SimpleNamespace(geom_size=M["geom_size"].copy()))
        env.filename = table[k]
        env.obj_size = "m"                               # what __init__ leaves (ENV:62) and the object schedule never updates
        sizes_sched.append(env._get_obj_size())
        env.obj_size = k[-1].lower()                     # what obj_shape_generator sets for [shape, size] (the obj_params hook)
        sizes_hook.append(env._get_obj_size())
        names.append(k)
    out["size_keys"] = np.array(names)
    out["size_schedule_path"] = np.array(sizes_sched, dtype=np.float64)
    out["size_obj_params_path"] = np.array(sizes_hook, dtype=np.float64)

    # ---- determine_hand_location
    Mh = mc.read_blob(scenarios.model_blob("CubeS"))
    rows = []
    for o in ("normal", "rotated", "top"):
        Rp = mc.quat_to_mat(scenarios.hand_quat_for(o)) @ mc.quat_to_mat(Mh["geom_quat"][1])
        for letter in "smb":
            env = new_env()
            data = SimpleNamespace(get_geom_xmat=lambda name, R=Rp: R.copy(), get_geom_xpos=lambda name: np.array([0.0, 0.18, 0.07]))
            env._sim = SimpleNamespace(data=data, model=SimpleNamespace(geom_size=Mh["geom_size"].copy()))
            env.filename, env.obj_size, env.orientation = "/kinova_description/j2s7s300_end_effector_v1_CubeS.xml", letter, o
            env._get_trans_mat_wrist_pose()
            rows.append([("normal", "rotated", "top").index(o), "smb".index(letter)] + list(env.determine_hand_location()))
    out["hand_location"] = np.array(rows, dtype=np.float64)          # class, letter, xloc, yloc, zloc, f1prox, f2prox, f3prox

    # ---- randomize_initial_pos_data_collection
    rows = []
    for k in ("BowlS", "RBowlB", "VaseM", "CubeS"):
        M = mc.read_blob(scenarios.model_blob(k))
        for o in ("normal", "rotated", "top", "side"):
            env = new_env()
            env._sim = SimpleNamespace(model=SimpleNamespace(geom_size=M["geom_size"].copy()))
            env.filename, env.obj_size = table[k], "m"
            np.random.seed(11)
            import io, contextlib
            with contextlib.redirect_stdout(io.StringIO()):
                xyz = env.randomize_initial_pos_data_collection(orientation=o)
            rows.append((k, o, [float(v) for v in xyz]))
    out["fallback_keys"] = np.array([f"{k}/{o}" for k, o, _ in rows])
    out["fallback_xyz"] = np.array([v for _, _, v in rows])

    # ---- sample_initial_object_hand_pos
    base = REF / "gym_kinova_gripper" / "envs" / "kinova_description" / "obj_hand_coords"
    rows = []
    for noise, with_noise, cls in (("no_noise", False, "Normal"), ("with_noise", True, "normal"), ("no_noise", False, "Top")):
        for shape in ("CubeS", "CylinderB"):
            f = base / noise / "train_coords" / cls / f"{shape}.txt"
            for region in (None, "left", "center", "target", "right", "origin"):
                env = new_env()
                np.random.seed(5)
                res = env.sample_initial_object_hand_pos(str(f), with_noise=with_noise, orient_idx=None, region=region)
                rows.append((f"{noise}/{cls}/{shape}/{region}", [float(v) for v in res[:6]], -1 if res[6] is None else int(res[6])))
    out["sample_keys"] = np.array([k for k, _, _ in rows])
    out["sample_xyz_hand"] = np.array([v for _, v, _ in rows])
    out["sample_idx"] = np.array([i for _, _, i in rows])
    np.savez_compressed(REPO / "tests" / "golden" / "reset_helpers.npz", **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
