#!/usr/bin/env python3
"""Study (test infrastructure; uses the CPU oracle): naive-controller success map of THIS physics over the cells of the
reference's recorded MuJoCo heat maps (tests/golden/mujoco_recorded.npz), side by side.
usage: python -m tests.studies.naive_heatmap [solver_iterations] [narrow_phase 0|1] > profiles/r03_naive_heatmap.txt"""
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))


def run_chunk(args):
    xs, ys, iters, narrow = args
    from kinovagrasping_amd import demonstrators
    from tests.oracle_vec import OracleVecSim, place_at_palm_xy
    torch.set_num_threads(1)
    sim = OracleVecSim(len(xs), "CubeS", solver_iterations=iters, rays=False, narrow_phase=narrow)
    q, hq, _ = place_at_palm_xy(sim, xs, ys)
    obs0 = sim.reset(torch.as_tensor(q), torch.as_tensor(hq))
    out = demonstrators.run_controller_episodes(sim, obs0.clone(), None, horizon=30, mode="naive", lift_rule="expert")
    return out["success"].numpy(), out["steps"].numpy()


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    narrow = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rec = np.load(ROOT / "tests" / "golden" / "mujoco_recorded.npz")
    hs, hf, hx, hy = rec["heat_success"], rec["heat_fail"], rec["heat_x"], rec["heat_y"]
    jj, ii = np.nonzero((hs > 0) | (hf > 0))
    ref = np.where(hs[jj, ii] > 0, hs[jj, ii], 100.0 - hf[jj, ii]) / 100.0
    chunks = np.array_split(np.arange(len(jj)), 32)
    with ProcessPoolExecutor(8) as ex:
        res = list(ex.map(run_chunk, [(hx[ii[c]], hy[jj[c]], iters, narrow) for c in chunks]))
    ours = np.concatenate([r[0] for r in res]).astype(bool)
    steps = np.concatenate([r[1] for r in res])
    ok, bad = ref >= 0.75, ref <= 0.25
    centre = bad & (np.abs(hx[ii]) < 0.04) & (hy[jj] < 0.055)
    corners = bad & ~centre
    print(f"naive controller, CubeS, normal hand pose, expert_data.py:746-804 loop; fp64 oracle, Newton <= {iters} iterations, narrow phase: "
          + ("GJK closest features + MPR on overlap (the product's)" if narrow == 0 else "MPR on margin-inflated hulls everywhere (MuJoCo 1.50's scheme; study mode)"))
    print(f"cells with recorded trials: {len(jj)}  (2 mm cells; one episode from each cell centre here)")
    print(f"recorded success cells (rate >= 75 %): {ok.sum():4d}  -> success here: {ours[ok].mean():.3f}")
    print(f"recorded failure cells, far corners  : {corners.sum():4d}  -> failure here: {1 - ours[corners].mean():.3f}")
    print(f"recorded failure cells, near-palm centre (|x| < 0.04, y < 0.055): {centre.sum():4d}  -> failure here: {1 - ours[centre].mean():.3f}")
    print(f"mixed cells (25-75 %): {(~ok & ~bad).sum()}")
    print(f"steps to done of successful episodes here: median {np.median(steps[ours]):.0f}, 5-95 % {np.percentile(steps[ours], 5):.0f}-{np.percentile(steps[ours], 95):.0f}"
          f"   (recorded demonstrations: 21-28)")
    grid_o = np.full(hs.shape, " ")
    grid_r = np.full(hs.shape, " ")
    for k in range(len(jj)):
        grid_o[jj[k], ii[k]] = "#" if ours[k] else "."
        grid_r[jj[k], ii[k]] = "#" if ref[k] >= 0.75 else ("." if ref[k] <= 0.25 else "+")
    print("\nrecorded MuJoCo (# success >= 75 %, . failure, + mixed)" + " " * 38 + "this physics (# success, . failure)")
    for j in range(hs.shape[0] - 1, -1, -1):
        if (grid_o[j] != " ").any():
            print(f"{hy[j]:5.3f} " + "".join(grid_r[j]) + "  |  " + "".join(grid_o[j]))
    print("      x = -0.09 ... 0.09 (palm frame, 2 mm per column)")


if __name__ == "__main__":
    main()
