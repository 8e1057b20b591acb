#!/usr/bin/env python3
"""Lift success of CONSTANT actions under the training rollout (exploration noise, check_grasp lift rule, scripted lift), 4096 CubeS envs, 4 episodes each."""
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kinovagrasping_amd import scenarios  # noqa: E402
from kinovagrasping_amd.ddpgfd import DDPGfD  # noqa: E402
from kinovagrasping_amd.rollout import RolloutEngine  # noqa: E402
from kinovagrasping_amd.sim import KinovaSim  # noqa: E402

n = 4096
dev = torch.device("cuda", 0)
q0, hq = scenarios.config2_states(n)
q0, hq = torch.as_tensor(q0), torch.as_tensor(hq)
sim = KinovaSim(n, "CubeS", auto_reset=True, horizon=30)
policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=dev)
for noise in (0.1, 0.0):
    for a in ([0.002, .5, .5, .5], [0.002, .66, .21, .66], [0.002, .69, .69, .40], [0.002, .79, .79, .79], [0.002, .3, .3, .3], [0.06, .5, .5, .5], [0.2, .5, .5, .5]):
        with torch.no_grad():
            policy.actor.l3.weight.zero_()
            policy.actor.l3.bias.copy_(torch.tensor([math.log((x / 0.8) / (1 - x / 0.8)) for x in a]))
        eng = RolloutEngine(sim, policy, None, expl_noise=noise)
        eng.start(sim.reset(q0, hq))
        lift = ep = 0
        tl = []
        for t in range(120):
            reward, done = eng.step()
            lift += int(((reward > 0) & done).sum()); ep += int(done.sum())
            tl.append(eng.lifting.float().mean().item())
        print(f"noise {noise}: constant action {a}: episodes {ep}, lift success {lift / max(1, ep):.3f}, envs in scripted lift (mean over steps) {sum(tl) / len(tl):.2f}")
sim.close()
