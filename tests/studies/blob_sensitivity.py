#!/usr/bin/env python3
"""Study (test infrastructure; CPU oracle): which model parameter moves the naive controller's outcome in the near-palm centre zone,
where recorded MuJoCo 1.50 fails and this physics lifts?  32 cells of that zone + 32 recorded-success cells, one
episode each, under edited model blobs.  usage: python -m tests.studies.blob_sensitivity > profiles/r03_blob_sensitivity.txt"""
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))


def run(args):
    name, edits, xs, ys, narrow = args
    from kinovagrasping_amd import demonstrators, scenarios
    from kinovagrasping_amd.model_compiler import blob_bytes, read_blob
    from oracle import ko_py as ko
    from tests.oracle_vec import OracleVecSim, place_at_palm_xy
    torch.set_num_threads(1)
    M = read_blob(scenarios.model_blob("CubeS"))
    for key, idx, val in edits:
        a = M[key].copy()
        if idx is None:
            a[...] = val
        else:
            a[idx] = val
        M[key] = a
    sim = OracleVecSim(len(xs), "CubeS", solver_iterations=100, rays=False, narrow_phase=narrow)
    sim.model = ko.OracleModel(blob_bytes(M))
    q, hq, _ = place_at_palm_xy(sim, xs, ys)
    obs0 = sim.reset(torch.as_tensor(q), torch.as_tensor(hq))
    out = demonstrators.run_controller_episodes(sim, obs0.clone(), None, horizon=30, mode="naive", lift_rule="expert")
    return name, out["success"].numpy(), out["steps"].numpy()


def main():
    rec = np.load(ROOT / "tests" / "golden" / "mujoco_recorded.npz")
    hs, hf, hx, hy = rec["heat_success"], rec["heat_fail"], rec["heat_x"], rec["heat_y"]
    jj, ii = np.nonzero((hs > 0) | (hf > 0))
    rate = np.where(hs[jj, ii] > 0, hs[jj, ii], 100.0 - hf[jj, ii]) / 100.0
    blob = np.flatnonzero((rate <= 0.25) & (np.abs(hx[ii]) < 0.04) & (hy[jj] < 0.055))
    good = np.flatnonzero(rate >= 0.75)
    rng = np.random.RandomState(0)
    sel = np.r_[rng.choice(blob, 32, replace=False), rng.choice(good, 32, replace=False)]
    xs, ys = hx[ii[sel]], hy[jj[sel]]
    S = slice(None)
    variants = [
        ("as compiled", []),
        ("impratio 1 (MuJoCo's default)", [("opt", 1, 1.0)]),
        ("impratio 25", [("opt", 1, 25.0)]),
        ("hand-object friction 0.5", [("pairs", (slice(1, 8), slice(2, 4)), 0.5)]),
        ("hand-object friction 0.3", [("pairs", (slice(1, 8), slice(2, 4)), 0.3)]),
        ("hand-object friction 0.15", [("pairs", (slice(1, 8), slice(2, 4)), 0.15)]),
        ("ground friction 1.0", [("pairs", (0, slice(2, 4)), 1.0)]),
        ("solref timeconst 0.05 (softer)", [("opt", 4, 0.05)]),
        ("solimp 0.5 / 0.95", [("opt", 6, 0.5)]),
        ("solimp 0.99 / 0.999 (stiffer)", [("opt", 6, 0.99), ("opt", 7, 0.999)]),
        ("finger servo kv 1.0", [("actuator", 3, 1.0)]),
        ("finger servo kv 0.5", [("actuator", 3, 0.5)]),
        ("slide servo kv 50", [("actuator", 0, 50.0)]),
        ("object mass 0.3 kg", [("body_mass", 9, 0.3)]),
        ("object mass 0.5 kg", [("body_mass", 9, 0.5)]),
        ("contact margin 0.005", [("pairs", (S, 4), 0.005)]),
        ("no hand-hand / hand-ground pairs", [("pairs", None, None)]),
    ]
    jobs = []
    for name, ed in variants:
        if name.startswith("no hand-hand"):
            continue
        jobs.append((name, ed, xs, ys, 0))
    jobs.append(("MuJoCo 1.50 narrow-phase scheme", [], xs, ys, 1))
    with ProcessPoolExecutor(8) as ex:
        res = list(ex.map(run, jobs))
    print("naive controller, CubeS, 32 cells of the near-palm centre zone (recorded MuJoCo: all fail) + 32 recorded-success cells; fp64 oracle")
    print(f"{'model variant':40s} {'zone cells lifted':>18s} {'success cells lifted':>22s} {'median steps (zone / success)':>32s}")
    for name, succ, steps in res:
        a, b = succ[:32], succ[32:]
        ms = lambda s, t: f"{np.median(t[s]):.0f}" if s.any() else "-"
        print(f"{name:40s} {int(a.sum()):>12d} / 32 {int(b.sum()):>16d} / 32 {ms(a, steps[:32]):>18s} / {ms(b, steps[32:])}")


if __name__ == "__main__":
    main()
