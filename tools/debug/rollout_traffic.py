#!/usr/bin/env python3
"""k_rollout alone (no learner) for counter passes: 4096 envs, CubeS, a fixed actor whose output bias closes the hand, 12 launches of 10 env-steps.
usage (GPU box): rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d DIR -- python3 tools/debug/rollout_traffic.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kinovagrasping_amd import scenarios  # noqa: E402
from kinovagrasping_amd.ddpgfd import DDPGfD  # noqa: E402
from kinovagrasping_amd.pipeline import AsyncTrainer  # noqa: E402
from kinovagrasping_amd.replay import DeviceEpisodeReplay  # noqa: E402
from kinovagrasping_amd.rollout import RolloutEngine  # noqa: E402
from kinovagrasping_amd.sim import KinovaSim  # noqa: E402

n = 4096
q0, hq = scenarios.config2_states(n)
sim = KinovaSim(n, "CubeS", horizon=30, auto_reset=True)
obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
torch.manual_seed(2)
policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=sim.device)
with torch.no_grad():
    policy.actor.l3.bias.copy_(torch.tensor([-6.0, 1.0, 0.8, 1.2]))
replay = DeviceEpisodeReplay(n, capacity=4 * n, horizon=30, device=sim.device)
eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
eng.start(obs0)
tr = AsyncTrainer(sim, policy, replay, eng, batch_episodes=64)
tr.capture()
for _ in range(12):
    tr.run(10, learn=False)
tr.flush(); torch.cuda.synchronize()
print(tr.counts(), flush=True)
sim.close()
