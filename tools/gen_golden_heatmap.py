#!/usr/bin/env python3
"""Dev-only: files the reference's heatmap bookkeeping writes for a given set of evaluation outcomes
(plotting_code/heatmap_coords.py: add_heatmap_coords 8-30, filter_heatmap_coords 34-66, save_coordinates 82-99)
-> tests/golden/heatmap.npz (inputs, relative file names, array contents, info text)."""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, "/root/reference/gym-kinova-gripper/plotting_code")
import heatmap_coords as hc  # noqa: E402


def main():
    rng = np.random.RandomState(11)
    n = 40
    xs, ys = rng.uniform(-0.09, 0.09, n), rng.uniform(-0.02, 0.08, n)
    orient = rng.choice(["normal", "rotated", "top"], n, p=[0.5, 0.3, 0.2])
    success = rng.rand(n) < 0.6
    success[orient == "top"] = True                      # an orientation without failures
    sc = {"x": [], "y": [], "orientation": []}
    fc = {"x": [], "y": [], "orientation": []}
    for i in range(n):
        ret = hc.add_heatmap_coords(sc, fc, orient[i], [xs[i], ys[i]], bool(success[i]))
        sc, fc = ret["success_coords"], ret["fail_coords"]
    out = {"x": xs, "y": ys, "orientation": orient, "success": success}
    with tempfile.TemporaryDirectory() as td:
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            hc.filter_heatmap_coords(sc, fc, 300, td)
        names = []
        for root, _, files in os.walk(td):
            for f in sorted(files):
                rel = os.path.relpath(os.path.join(root, f), td)
                names.append(rel)
                if f.endswith(".npy"):
                    out["file:" + rel] = np.load(os.path.join(root, f))
                else:
                    out["text:" + rel] = np.array(open(os.path.join(root, f)).read().replace(td, "<DIR>"))
        out["names"] = np.array(sorted(names))
    dst = REPO / "tests" / "golden" / "heatmap.npz"
    np.savez_compressed(dst, **out)
    print("wrote", dst, len(names), "files")


if __name__ == "__main__":
    main()
