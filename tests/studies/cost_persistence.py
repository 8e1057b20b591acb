#!/usr/bin/env python3
"""Study (needs a GPU): is an env's cost a property of the ENV (each env of config 2 / 3 resets to its own start row) or of the episode phase?

BASELINE config 3 (4096 envs, CubeS, 256-256, one update per env-step) is trained for 900 env-steps in lock step; over the next 120 env-steps
(four episodes) the contact count of every env after every env-step is recorded - the proxy for an env-step's cost (Newton iterations and live hull
pairs both follow it).  Reported: the share of the variance of the per-step contact count that belongs to the env (between-env variance of the
120-step means), the correlation of an env's episode totals between consecutive episodes, and what a 16-env group pays (the maximum over its
envs) against the mean env.
usage (GPU box): python -m tests.studies.cost_persistence > profiles/r03_cost_persistence.txt"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from kinovagrasping_amd import scenarios  # noqa: E402
from kinovagrasping_amd.ddpgfd import DDPGfD  # noqa: E402
from kinovagrasping_amd.pipeline import GraphedTrainer  # noqa: E402
from kinovagrasping_amd.replay import DeviceEpisodeReplay  # noqa: E402
from kinovagrasping_amd.rollout import RolloutEngine  # noqa: E402
from kinovagrasping_amd.sim import KinovaSim  # noqa: E402

n, T = 4096, 120
q0, hq = scenarios.config2_states(n)
sim = KinovaSim(n, "CubeS", horizon=30, auto_reset=True)
obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
torch.manual_seed(2)
policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=sim.device, capturable=True)
replay = DeviceEpisodeReplay(n, capacity=4 * n, horizon=30, device=sim.device)
eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
eng.start(obs0)
tr = GraphedTrainer(sim, policy, replay, eng, batch_episodes=64)
tr.capture()
for _ in range(900):
    tr.step()
tr.flush()
ncon = np.zeros((T, n))
clock = np.zeros((T, n), dtype=np.int64)
for t in range(T):
    tr.step()
    tr.flush()
    ncon[t] = sim.get_state()["ncon"].cpu().numpy()
    clock[t] = eng.t.cpu().numpy()
sim.close()

env_mean = ncon.mean(0)
print(f"config 3 after {tr.updates} updates, {T} env-steps recorded; contacts per env after an env-step: mean {ncon.mean():.2f}, max {ncon.max():.0f}")
print(f"variance of the per-step contact count: total {ncon.var():.3f}; between envs (variance of the {T}-step means) {env_mean.var():.3f} "
      f"= {env_mean.var() / ncon.var():.2f} of the total")
quart = ncon.reshape(4, T // 4, n).sum(1)                     # four consecutive 30-step windows (~ one episode each)
cc = [np.corrcoef(quart[i], quart[i + 1])[0, 1] for i in range(3)]
print(f"an env's 30-step contact totals, correlation between consecutive windows: {cc[0]:.2f} {cc[1]:.2f} {cc[2]:.2f}")
order = np.argsort(-env_mean)
top = set(order[: n // 10].tolist())
for i in range(4):
    ti = set(np.argsort(-quart[i])[: n // 10].tolist())
    print(f"  window {i}: the top-10 % envs of the whole record hold {len(top & ti) / (n // 10):.2f} of this window's top 10 %")
grp = ncon.reshape(T, n // 16, 16)
print(f"what a 16-env group pays per env-step (max over its envs) {grp.max(2).mean():.2f} contacts against the mean env {ncon.mean():.2f}; "
      f"the busiest group's {T}-step mean of that maximum {grp.max(2).mean(0).max():.2f}, the median group's {np.median(grp.max(2).mean(0)):.2f}, the calmest {grp.max(2).mean(0).min():.2f}")
ph = np.array([ncon[clock == k].mean() if (clock == k).any() else np.nan for k in range(30)])
print("mean contacts by the episode clock after the step (0 = just reset):", " ".join(f"{x:.1f}" for x in ph))
