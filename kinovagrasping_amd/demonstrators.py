"""Scripted demonstrators run batched on the GPU (first part).

NaiveController (gym-kinova-gripper/expert_data.py:596-607 and the "naive" branch of get_action,
expert_data.py:610-671): close all three fingers at constant_velocity = 0.5 until check_grasp fires
(after >= 6 steps), then lift with [wrist 0.6, fingers 0.5].  The episodes fill an expert replay that
DDPGfD.train_batch mixes in at 30 % (DDPGfD.py:232-254).  The position-dependent "nudge" PID controller
(expert_data.py:318-537) is the next row and is not implemented yet.
"""
from __future__ import annotations

import torch

from .rollout import SKIP_NUM_TS, check_grasp

VELOCITIES = {"constant_velocity": 0.5, "min_velocity": 0.5, "max_velocity": 0.8, "finger_lift_velocity": 0.5,
              "wrist_lift_velocity": 0.6}     # expert_data.py:617


def naive_action(lift_check: torch.Tensor) -> torch.Tensor:
    """Batched NaiveController: lift_check [N] bool -> actions [N, 4]."""
    v = VELOCITIES
    close = torch.tensor([0.0, v["constant_velocity"], v["constant_velocity"], v["constant_velocity"]], device=lift_check.device)
    lift = torch.tensor([v["wrist_lift_velocity"], v["finger_lift_velocity"], v["finger_lift_velocity"], v["finger_lift_velocity"]],
                        device=lift_check.device)
    return torch.where(lift_check.unsqueeze(1), lift, close)


@torch.no_grad()
def run_naive_episodes(sim, obs0: torch.Tensor, replay=None, horizon: int = 30):
    """One episode per env with the naive controller (sim must have auto_reset=False or horizon >= `horizon`).
    Returns dict(success [N] bool, steps [N], total_reward [N]).  Transitions go to `replay` (all steps are
    stored for demonstrations, expert_data.py:746-804)."""
    n, dev = sim.n_envs, sim.device
    obs = obs0.clone()
    prev = None
    ready = torch.zeros(n, dtype=torch.bool, device=dev)
    alive = torch.ones(n, dtype=torch.bool, device=dev)
    success = torch.zeros(n, dtype=torch.bool, device=dev)
    steps = torch.zeros(n, dtype=torch.long, device=dev)
    total = torch.zeros(n, device=dev)
    for t in range(horizon):
        if prev is not None and t + 1 >= SKIP_NUM_TS:
            ready |= check_grasp(prev[:, 9:17], obs[:, 9:17]) & alive
        action = naive_action(ready)
        state = obs
        nobs, reward, done, info = sim.step(action.t().contiguous())
        done_b = (done != 0) & alive
        nxt = torch.where(done_b.unsqueeze(1), sim.final_obs, nobs) if sim.cfg.auto_reset else nobs
        if replay is not None:
            replay.add(state, action, nxt, reward, done_b | (t == horizon - 1), store_mask=alive)
        total += torch.where(alive, reward, torch.zeros_like(reward))
        steps += alive.long()
        success |= done_b & (info[2] > 0)
        alive &= ~done_b
        prev, obs = state, nobs.clone()
        if not alive.any():
            break
    if replay is not None:
        replay.end_episodes(torch.ones(n, dtype=torch.bool, device=dev))
    return {"success": success, "steps": steps, "total_reward": total}
