"""Batched policy evaluation: the counterpart of eval_policy in
gym-kinova-gripper/main_DDPGfD.py:130-272 with one evaluation episode per env.

As in the reference: the deterministic policy acts (no exploration noise) until check_grasp - after >= 6 steps -
latches the lift (main_DDPGfD.py:179-196), then the scripted lift action [0.6, 0.5, 0.5, 0.5] is repeated until the
episode ends (eval_lift_hand, main_DDPGfD.py:293-307); an episode is a success when a step reward exceeds 25
(the 50-point lift reward, main_DDPGfD.py:222-223); start coordinates of the object in the palm frame are
collected per outcome for the success / fail heatmaps (add_heatmap_coords, main_DDPGfD.py:310-330).
"""
from __future__ import annotations

import torch

from .rollout import LIFT_ACTION, SKIP_NUM_TS, check_grasp


@torch.no_grad()
def eval_policy(sim, policy, obs0: torch.Tensor, horizon: int = 30, orientation: str = "normal"):
    """One evaluation episode per env of `sim` (auto_reset off, or horizon <= the sim's).  obs0 [N, 82] from the reset.
    Returns the reference's result dict: avg_reward, avg_rewards{total,finger,grasp,lift}, all_ep_reward_values,
    num_success, success_coords / fail_coords {x, y, orientation} (object start position in the palm frame), plus
    `success` [N] bool and `steps` [N] tensors."""
    n, dev = sim.n_envs, sim.device
    obs = obs0.clone()
    start_xy = obs0[:, 21:23].clone()                       # Tfw . object position (main_DDPGfD.py:166-171) = obs[21:24]
    prev = None
    ready = torch.zeros(n, dtype=torch.bool, device=dev)
    alive = torch.ones(n, dtype=torch.bool, device=dev)
    success = torch.zeros(n, dtype=torch.bool, device=dev)
    steps = torch.zeros(n, dtype=torch.long, device=dev)
    totals = torch.zeros(4, n, device=dev)                  # total, finger, grasp, lift reward per episode
    lift = torch.tensor(LIFT_ACTION, device=dev).expand(n, 4)
    for t in range(horizon):
        if prev is not None and t + 1 >= SKIP_NUM_TS:
            ready |= check_grasp(prev[:, 9:17], obs[:, 9:17]) & alive
        action = torch.where(ready.unsqueeze(1), lift, policy.select_action(obs))
        state = obs
        nobs, reward, done, info = sim.step(action.t().contiguous())
        live = alive.float()
        totals[0] += reward * live
        totals[1:] += info * live
        steps += alive.long()
        done_b = (done != 0) & alive
        success |= alive & (reward > 25)
        alive &= ~done_b
        prev, obs = state, nobs.clone()
    xs, ys = start_xy[:, 0].cpu().numpy(), start_xy[:, 1].cpu().numpy()
    ok = success.cpu().numpy()
    coords = lambda m: {"x": xs[m].tolist(), "y": ys[m].tolist(), "orientation": [orientation] * int(m.sum())}
    tot = totals.cpu().numpy()
    names = ("total_reward", "finger_reward", "grasp_reward", "lift_reward")
    return {"avg_reward": float(tot[0].mean()), "avg_rewards": {k: float(tot[i].mean()) for i, k in enumerate(names)},
            "all_ep_reward_values": {k: tot[i].tolist() for i, k in enumerate(names)}, "num_success": int(ok.sum()),
            "success_coords": coords(ok), "fail_coords": coords(~ok), "success": success, "steps": steps}
