"""Dev tool (diagnostic build -DKS_RAY_COUNT): node visits of every (ray, geom) walk of k_rays over a closing grasp.
usage: KS_LIB=<count build> python tools/ray_visits.py"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from kinovagrasping_amd import scenarios
from kinovagrasping_amd.sim import KinovaSim

n = 4096
q0, hq = scenarios.config2_states(n)
sim = KinovaSim(n, "CubeS", auto_reset=False, horizon=30, contact_tap=False)
sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
act = torch.tensor([0.0, 0.5, 0.5, 0.5], device="cuda").repeat(n, 1).t().contiguous()
names = ["palm", "f1p", "f1d", "f2p", "f2d", "f3p", "f3d", "object"]
for t in range(16):
    sim.step(act)
    if t in (0, 8, 15):
        c = sim.get_state(contacts=True)["contact"].reshape(-1, n)[:136].cpu().numpy().reshape(17, 8, n)
        print(f"step {t}: visits per (ray, geom) walk: mean {c.mean():.1f}  nonzero {100 * (c > 0).mean():.0f}%  mean over walks {c[c > 0].mean():.1f}  p99 {np.percentile(c[c > 0], 99):.0f}  max {c.max():.0f}")
        wave = c.transpose(0, 2, 1).reshape(17, n // 8, 64)          # a wave = 8 envs x 8 geoms of one ray
        wm = wave.max(2)
        print(f"   per wave: mean of max {wm.mean():.1f}  p90 {np.percentile(wm, 90):.0f}  max {wm.max():.0f}")
        print("   mean visits by geom:", " ".join(f"{names[g]} {c[:, g].mean():.1f}" for g in range(8)))
        print("   max visits by geom: ", " ".join(f"{names[g]} {c[:, g].max():.0f}" for g in range(8)))
        print("   mean of wave max by ray:", " ".join(f"{wm[r].mean():.0f}" for r in range(17)))
