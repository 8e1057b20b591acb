#!/usr/bin/env python3
"""Dev-only (GPU box): free-running long-horizon parity of the MULTI-GEOM objects from START STATES A RESET PRODUCES (the study
of rounds 4-5 placed the main piece in the hand, where the bowls overlap the closing fingers by 1 - 2.5 cm - a state no reset produces): six
draws per object of KinovaGripperVecEnv.reset (no-noise start table of the object or, where the reference ships none, its empty-file rule; the
reset's 5 cm correction applied), closing grasp + lift script, 210 substeps; fp32 and fp64 instantiations of libkinova_sim_mg.so against the fp64
oracle, through ks_substep (tests/studies/long_horizon.py) and, fp32, through ks_step (tests/studies/long_horizon_envstep.py)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kinovagrasping_amd import scenarios
from kinovagrasping_amd.vec_env import KinovaGripperVecEnv
from tests.studies import long_horizon as lh
from tests.studies import long_horizon_envstep as le

n_sub, T, per = 210, 14, 6
script = np.array([[0.0, 0.6, 0.5, 0.7]] * 9 + [[0.6, 0.5, 0.5, 0.5]] * (T - 9))
shapes = ("BottleS", "BottleB", "TBottleS", "TBottleM", "BowlS", "BowlB", "RBowlS", "RBowlM")
starts = {}
for sh in shapes:
    env = KinovaGripperVecEnv(per, sh, seed=11, host_only=True)
    st = env.reset([sh], "normal", with_noise=False)
    starts[sh] = (st["qpos"], st["hand_quat"])
tot = {}
for prec, path in ((32, "ks_substep"), (32, "ks_step"), (64, "ks_substep")):
    print(f"fp{prec} kernels of libkinova_sim_mg.so through {path} vs fp64 oracle, free running {n_sub} substeps (metric: |dqpos|_inf / max(1e-3, |qpos|_inf))")
    tot[(prec, path)] = 0
    for sh in shapes:
        q0, hq = starts[sh]
        acts = np.repeat(script[:, :, None], per, 2)
        tol = 1e-4 if prec == 32 else 1e-9
        if path == "ks_substep":
            res = lh.run_batch(sh, q0, hq, acts, n_sub, precision=prec)
            r, fb = res["rel"][199], lh.first_bad(res["rel"])
            extra = f"   first substep beyond 1e-4: {sorted(int(x) for x in fb[fb >= 0])}   lifted at the end {int((res['phase'][-1] == 3).sum())}/{per}"
        else:
            res = le.run_batch(sh, q0, hq, acts, precision=prec)
            r, extra = res["rel"][13], ""
        tot[(prec, path)] += int((r <= tol).sum())
        print(f"  {sh:9s} substep {200 if path == 'ks_substep' else 210}: within {tol:g} {int((r <= tol).sum())}/{len(r)}   median {np.median(r):.1e}  max {r.max():.1e}{extra}   "
              f"status {sorted(set(res['status'].tolist()))}", flush=True)
print("totals:", {f"fp{k[0]} {k[1]}": f"{v} of {per * len(shapes)}" for k, v in tot.items()})
