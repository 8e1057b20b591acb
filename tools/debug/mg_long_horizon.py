#!/usr/bin/env python3
"""Dev-only (GPU box): free-running long-horizon parity of the MULTI-GEOM objects - the main piece placed in the hand at body height 0 (the reference's
reset: see scenarios.reset_body_position), 6 starts per object, closing grasp + lift script, 210 substeps;
fp32 and fp64 instantiations of libkinova_sim_mg.so against the fp64 oracle (tests/studies/long_horizon.py machinery)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kinovagrasping_amd import model_compiler as mc, scenarios
from tests.studies import long_horizon as lh

n_sub, T = 210, 14
script = np.array([[0.0, 0.6, 0.5, 0.7]] * 9 + [[0.6, 0.5, 0.5, 0.5]] * (T - 9))
for prec in (32, 64):
    print(f"fp{prec} kernels of libkinova_sim_mg.so vs fp64 oracle, free running {n_sub} substeps (metric: |dqpos|_inf / max(1e-3, |qpos|_inf))")
    for sh in ("BottleS", "BottleB", "TBottleS", "TBottleM", "BowlS", "BowlB", "RBowlS", "RBowlM"):
        g = mc.read_blob(scenarios.model_blob(sh))["geom_pos"][8]
        offs = [(0, 0), (0.02, 0), (-0.02, 0.005), (0.01, -0.01), (0.03, 0.01), (-0.03, -0.005)]
        q0 = np.zeros((16, len(offs))); q0[12] = 1
        for i, (dx, dy) in enumerate(offs):
            q0[9:12, i] = [-g[0] + dx, -g[1] + dy, 0.0]
        hq = np.repeat(scenarios.hand_quat_for("normal")[:, None], len(offs), 1)
        acts = np.repeat(script[:, :, None], len(offs), 2)
        res = lh.run_batch(sh, q0, hq, acts, n_sub, precision=prec)
        r = res["rel"][199]
        fb = lh.first_bad(res["rel"])
        tol = 1e-4 if prec == 32 else 1e-9
        lifted = int((res["phase"][-1] == 3).sum())
        print(f"  {sh:9s} substep 200: within {tol:g} {int((r <= tol).sum())}/{len(r)}   median {np.median(r):.1e}  max {r.max():.1e}   first substep beyond 1e-4: "
              f"{sorted(int(x) for x in fb[fb >= 0])}   lifted at the end {lifted}/{len(r)}   status {sorted(set(res['status'].tolist()))}", flush=True)
