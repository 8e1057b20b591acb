#!/usr/bin/env python3
"""Dev-only: known answers of the reference's staged-experiment tables (main_DDPGfD.py: get_experiment_info 624-671,
get_exp_input 710-738, and the shape / size lists of the `experiment` mode, 1270-1300) -> tests/golden/curriculum.json.
The two functions are pure; they are compiled from the reference's file in an empty namespace (importing the whole
driver would need gym, MuJoCo and its argument parser)."""
import ast
import contextlib
import io
import json
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
SRC = Path("/root/reference/gym-kinova-gripper/main_DDPGfD.py")


def main():
    tree = ast.parse(SRC.read_text())
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("get_experiment_info", "get_exp_input")]
    assert len(wanted) == 2
    ns = {}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), str(SRC), "exec"), ns)
    out = {"info": {}, "input": []}
    with contextlib.redirect_stdout(io.StringIO()):
        for exp_num in range(0, 18):
            try:
                out["info"][str(exp_num)] = list(ns["get_experiment_info"](exp_num))
            except Exception as e:                       # the reference fails outside 1..6 (its `stage3` table is commented out)
                out["info"][str(exp_num)] = type(e).__name__
        train_shapes = ["Cube", "Cylinder", "Cube45", "Vase2", "Bottle", "Bowl", "TBottle"]      # main_DDPGfD.py:1281-1282
        combos = [(train_shapes, ["S", "B"]), (["Vase1", "RBowl"], ["M"])]                        # train / test lists, 1270-1282
        names = ["sizes", "shapes", "orientations", "sizes_shapes_orientations", "shapes_sizes_orientations", "orientations_sizes_shapes",
                 "kitchen_sink", "pretrain_policy", "sizes_shapes", "shapes_orientations", "sizes_orientations"]
        for shapes, sizes in combos:
            for name in names:
                req, ori = ns["get_exp_input"](name, shapes, sizes)
                out["input"].append({"exp_name": name, "shapes": shapes, "sizes": sizes, "requested_shapes": req, "orientation": ori})
    dst = REPO / "tests" / "golden" / "curriculum.json"
    dst.write_text(json.dumps(out, indent=1))
    print("wrote", dst, out["info"])


if __name__ == "__main__":
    main()
