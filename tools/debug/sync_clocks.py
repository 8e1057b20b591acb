#!/usr/bin/env python3
"""Experiment: what would 16-env groups of envs at the SAME episode clock be worth?  Free-running rollout + learner (config 3), trained for 900
env-steps; then (a) 60 env-steps as they are (clocks spread by the early lifts), (b) every env restarted at once (clocks equal: every group is
homogeneous, and a launch of 30 env-steps covers a whole episode for every group) and the next 30 / 60 env-steps timed.
usage (GPU box): python tools/debug/sync_clocks.py"""
import sys
import time
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
q0, hq = torch.as_tensor(q0), torch.as_tensor(hq)
sim = KinovaSim(n, "CubeS", horizon=30, auto_reset=True)
obs0 = sim.reset(q0, hq)
torch.manual_seed(2)
policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=sim.device)
replay = DeviceEpisodeReplay(n, capacity=4 * n, horizon=30, device=sim.device)
eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
eng.start(obs0)
tr = AsyncTrainer(sim, policy, replay, eng, batch_episodes=64)
tr.capture()
tr.run(36, learn=False); tr.flush()
for _ in range(15):
    tr.run(60)
tr.flush(); torch.cuda.synchronize()


def timed(k, reps, learn=True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        tr.run(k, learn=learn)
    tr.flush(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * k) * 1e3


def hist():
    return torch.bincount(eng.t.clamp(0, 29), minlength=30).cpu().tolist()


for learn in (True, False):
    print(f"learner {'beside' if learn else 'off'}: clocks as they are {hist()}")
    print(f"   60 env-steps in launches of 30: {timed(30, 2, learn):.4f} ms per env-step")
    for rep in range(2):
        tr.flush(); torch.cuda.synchronize()
        eng.start(sim.reset(q0, hq))                    # every env back to its start row, clock 0; open episodes dropped
        replay.a_len.zero_(); replay.pub_len.zero_(); eng.lifting.zero_()
        torch.cuda.synchronize()
        a = timed(30, 1, learn)
        h = hist()
        b = timed(30, 1, learn)
        print(f"   all clocks equal: first 30 env-steps {a:.4f} ms per env-step, next 30 {b:.4f} (clocks after the first 30: {h})")
sim.close()
