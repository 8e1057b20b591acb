#!/usr/bin/env python3
"""GPU box: fp64 kernels vs the oracle, lock step from the ORACLE's state every substep (so that one-step differences are not
compounded): for a shape and the 14-shape test's starts, report every substep whose qpos differs by more than 1e-11 with the two
contact lists.  usage: python tools/debug/fp64_first_diff.py CylinderB [max_reports]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kinovagrasping_amd import scenarios  # noqa: E402
from kinovagrasping_amd.sim import SOLVER_ITERATIONS, KinovaSim  # noqa: E402
from oracle import ko_py as ko  # noqa: E402
from tests.test_gpu_obs_contacts import POSES, ctrl_of, pose_start  # noqa: E402

np.set_printoptions(precision=9, suppress=True, linewidth=220)
sh = sys.argv[1]
max_rep = int(sys.argv[2]) if len(sys.argv) > 2 else 4
act = np.array([0.0, 0.6, 0.5, 0.7])
q0s, hqs = [], []
for ori in POSES:
    for k in range(4):
        q0, hq = pose_start(sh, ori, 17 + 997 * k)
        q0s.append(q0); hqs.append(hq)
n = len(q0s)
model = ko.OracleModel(scenarios.model_blob(sh))
orc = [ko.OracleSim(model, hqs[i], solver_iterations=SOLVER_ITERATIONS) for i in range(n)]
for i, o in enumerate(orc):
    o.s.rays_enabled = 0
    o.env_reset(q0s[i])
sim = KinovaSim(n, sh, horizon=0, precision=64, contact_tap=True, solver_iterations=SOLVER_ITERATIONS)
sim.reset(torch.as_tensor(np.stack(q0s, 1)), torch.as_tensor(np.stack(hqs, 1)))
reports = 0
for sub in range(60):
    ctrl = np.stack([ctrl_of(o, act) for o in orc], 1) if sub % 15 == 0 else ctrl
    before = [(o.view("qpos").copy(), o.view("qvel").copy(), o.view("qacc_warmstart").copy()) for o in orc]
    sim.set_state(torch.as_tensor(np.stack([b[0] for b in before], 1)), torch.as_tensor(np.stack([b[1] for b in before], 1)),
                  torch.as_tensor(np.stack([b[2] for b in before], 1)))
    sim.substep(torch.as_tensor(ctrl))
    st = sim.get_state(contacts=True)
    qg = st["qpos"].cpu().numpy()
    for i, o in enumerate(orc):
        o.step(ctrl[:, i])
        e = np.abs(qg[:, i] - o.view("qpos")).max()
        if e > 1e-11 and reports < max_rep:
            reports += 1
            nc = int(st["ncon"][i])
            con = st["contact"][:, :, i].cpu().numpy()
            print(f"{sh} env {i} substep {sub}: |dqpos| {e:.3e}; ncon gpu {nc} oracle {o.s.ncon}; newton iters {o.s.newton_iters_used}")
            for k, c in enumerate(o.contacts()):
                print("   oracle", c["geom1"], c["geom2"], "%.12e" % c["dist"], c["pos"], c["frame"][:3])
                if k < nc:
                    print("   gpu   ", int(con[k][8]) & 255, "%.12e" % con[k][6], con[k][:3], con[k][3:6])
print("reports", reports)
