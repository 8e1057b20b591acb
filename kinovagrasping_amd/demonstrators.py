"""Scripted demonstrators run batched on the GPU (first part).

NaiveController (gym-kinova-gripper/expert_data.py:596-607 and the "naive" branch of get_action,
expert_data.py:610-671): close all three fingers at constant_velocity = 0.5 until check_grasp fires
(after >= 6 steps), then lift with [wrist 0.6, fingers 0.5].  The episodes fill an expert replay that
DDPGfD.train_batch mixes in at 30 % (DDPGfD.py:232-254).

Position-dependent "nudge" controller (ExpertPIDController.PDController + PID, expert_data.py:318-537) and the
combined controller (get_action, expert_data.py:610-671), as the reference code BEHAVES (pinned by
tests/golden/controllers.npz, generated from the reference itself):
  * the PD controller picks a branch from the object's INITIAL palm-frame x (|x0| <= 0.03 centre, x0 > 0 right =
    two-finger side, x0 < 0 left = thumb side) and, in the side branches, from how far the object/palm alignment
    obs[81] has moved from its initial value (pre / post contact) and from 1;
  * check_vel_in_range then forces EVERY finger velocity into [min_velocity, max_velocity] = [0.5, 0.8] - its
    "leave 0 / lift values alone" test is a tautology - so zeros and halved lift velocities become 0.5;
  * the combined mode uses the CURRENT x: |x| > 0.04 PD controller, 0.02 <= |x| <= 0.04 "interpolation" =
    np.interp(arange(1, 4), naive[1:3], expert[1:3]) whose sample points all lie right of naive[1:3] = (0.5, 0.5),
    i.e. all three fingers take the PD controller's finger-2 velocity, else naive;
  * the wrist is wrist_lift_velocity when the lift flag is up, else 0.
"""
from __future__ import annotations

import torch

from .rollout import SKIP_NUM_TS, check_grasp

VELOCITIES = {"constant_velocity": 0.5, "min_velocity": 0.5, "max_velocity": 0.8, "finger_lift_velocity": 0.5,
              "wrist_lift_velocity": 0.6}     # expert_data.py:617


def naive_action(lift_check: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """Batched NaiveController: lift_check [N] bool -> actions [N, 4]."""
    v = VELOCITIES
    close = torch.tensor([0.0, v["constant_velocity"], v["constant_velocity"], v["constant_velocity"]], device=lift_check.device, dtype=dtype)
    lift = torch.tensor([v["wrist_lift_velocity"], v["finger_lift_velocity"], v["finger_lift_velocity"], v["finger_lift_velocity"]],
                        device=lift_check.device, dtype=dtype)
    return torch.where(lift_check.unsqueeze(1), lift, close)


def pd_controller_action(obs: torch.Tensor, init_x: torch.Tensor, init_dot: torch.Tensor, lift_check: torch.Tensor) -> torch.Tensor:
    """Batched ExpertPIDController.PDController (expert_data.py:318-537): obs [N, 82], init_x / init_dot [N] = obs[:, 21] /
    obs[:, 81] at the start of the episode, lift_check [N] bool -> actions [N, 4] (wrist, f1, f2, f3)."""
    v = VELOCITIES
    const, lift_w, lift_f, vmin, vmax = v["constant_velocity"], v["wrist_lift_velocity"], v["finger_lift_velocity"], v["min_velocity"], v["max_velocity"]
    dot, d78, d79 = obs[:, 81], obs[:, 78], obs[:, 79]
    k = 1.0 + 1.0 / 15.0                                    # kp + kd / sampling_time  (PID.velocity / touch_vel)
    pid_vel = ((1.0 - dot) * k / 1.25 * 0.3).clamp(min=0.05)
    touch1, touch2 = (dot - d78) * k, (dot - d79) * k
    zero, full = torch.zeros_like(dot), torch.full_like(dot, const)
    moved = (dot - init_dot).abs() > 0.01                   # centre: "> 0.01"; sides: pre-contact is "< 0.01"
    pre = (dot - init_dot).abs() < 0.01
    far = (1.0 - dot).abs() > 0.01
    lift = lift_check
    # centre
    c_f1 = torch.where(lift, torch.full_like(dot, lift_f / 2), full)
    c_f23 = torch.where(lift, torch.full_like(dot, lift_f), torch.where(moved, full / 2, full))
    # right: fingers 2, 3 push first
    r_f1 = torch.where(pre, zero, torch.where(lift, torch.full_like(dot, lift_f / 2), torch.where(far, torch.full_like(dot, vmin), touch1)))
    r_f23 = torch.where(pre, touch2, torch.where(lift, torch.full_like(dot, lift_f), torch.where(far, pid_vel, zero)))
    # left: finger 1 pushes first
    l_f1 = torch.where(pre, touch1, torch.where(lift, torch.full_like(dot, lift_f / 2), torch.where(far, pid_vel, zero)))
    l_f23 = torch.where(pre, zero, torch.where(lift, torch.full_like(dot, lift_f), torch.where(far, torch.full_like(dot, vmin), touch2)))
    centre, right = init_x.abs() <= 0.03, init_x > 0.0
    f1 = torch.where(centre, c_f1, torch.where(right, r_f1, l_f1))
    f23 = torch.where(centre, c_f23, torch.where(right, r_f23, l_f23))
    fingers = torch.stack([f1, f23, f23], 1).clamp(vmin, vmax)       # check_vel_in_range (expert_data.py:540-551)
    wrist = torch.where(lift, torch.full_like(dot, lift_w), zero)
    return torch.cat([wrist.unsqueeze(1), fingers], 1)


def controller_action(mode: str, obs: torch.Tensor, init_x: torch.Tensor, init_dot: torch.Tensor, lift_check: torch.Tensor) -> torch.Tensor:
    """Batched expert_data.get_action (expert_data.py:610-671) for mode "naive" | "position-dependent" | "combined"."""
    naive = naive_action(lift_check, obs.dtype)
    if mode == "naive":
        return naive
    pd = pd_controller_action(obs, init_x, init_dot, lift_check)
    if mode == "position-dependent":
        return pd
    if mode != "combined":
        raise ValueError(mode)
    x = obs[:, 21]
    outer = (x < -0.04) | (x > 0.04)
    band = ((x >= -0.04) & (x <= -0.02)) | ((x >= 0.02) & (x <= 0.04))
    interp = torch.cat([pd[:, :1], pd[:, 2:3].expand(-1, 3)], 1)      # np.interp right of its sample points: expert finger 2
    out = torch.where(outer.unsqueeze(1), pd, torch.where(band.unsqueeze(1), interp, naive))
    out[:, 0] = torch.where(lift_check, torch.full_like(x, VELOCITIES["wrist_lift_velocity"]), torch.zeros_like(x))
    return out


MIN_LIFT_TIMESTEPS = 10        # expert_data.py:762


@torch.no_grad()
def run_controller_episodes(sim, obs0: torch.Tensor, replay=None, horizon: int = 30, mode: str = "combined", lift_rule: str = "expert"):
    """One episode per env with a scripted demonstrator.  Returns dict(success [N] bool, steps [N], total_reward [N]).

    lift_rule "expert" (default) is the demonstration loop of expert_data.py:746-804 as it behaves: prev_obs is first set at
    the END of step 1 (`if total_steps > 0: prev_obs = obs`, :798), so check_grasp runs from step 2 on; every hit counts
    (`num_good_grasps`, never reset); the lift flag is `total_steps > min_lift_timesteps (10) and num_good_grasps >= 1`
    (:762-772); transitions are stored only while NOT lifting (:789-790) and an episode that ends during the lift
    overwrites its last stored transition with the outcome (`replay_buffer.replace`, :792-793, utils.py:309-343).
    lift_rule "train" is the training loop's rule (main_DDPGfD.py:418-439): check_grasp from the 6th step on, latched; every
    transition stored (what round 2 used for demonstrations too)."""
    if lift_rule not in ("expert", "train"):
        raise ValueError(lift_rule)
    n, dev = sim.n_envs, sim.device
    obs = obs0.clone()
    init_x, init_dot = obs[:, 21].clone(), obs[:, 81].clone()
    prev = None
    ready = torch.zeros(n, dtype=torch.bool, device=dev)
    alive = torch.ones(n, dtype=torch.bool, device=dev)
    success = torch.zeros(n, dtype=torch.bool, device=dev)
    steps = torch.zeros(n, dtype=torch.long, device=dev)
    total = torch.zeros(n, device=dev)
    for t in range(horizon):
        if lift_rule == "expert":
            if prev is not None and t >= 2:
                ready |= check_grasp(prev[:, 9:17], obs[:, 9:17]) & alive
            lift = ready & (t > MIN_LIFT_TIMESTEPS)
        else:
            if prev is not None and t + 1 >= SKIP_NUM_TS:
                ready |= check_grasp(prev[:, 9:17], obs[:, 9:17]) & alive
            lift = ready
        action = controller_action(mode, obs, init_x, init_dot, lift).to(obs.dtype)
        state = obs
        nobs, reward, done, info = sim.step(action.t().contiguous())
        done_b = (done != 0) & alive
        nxt = torch.where(done_b.unsqueeze(1), sim.final_obs, nobs) if sim.cfg.auto_reset else nobs
        if replay is not None:
            if lift_rule == "expert":
                replay.add(state, action, nxt, reward, done_b | (t == horizon - 1), store_mask=alive & ~lift)
                replay.replace_last(alive & lift & done_b, reward)
            else:
                replay.add(state, action, nxt, reward, done_b | (t == horizon - 1), store_mask=alive)
        total += torch.where(alive, reward, torch.zeros_like(reward))
        steps += alive.long()
        success |= done_b & (info[2] > 0)
        alive &= ~done_b
        prev, obs = state, nobs.clone()
    if replay is not None:
        replay.end_episodes(torch.ones(n, dtype=torch.bool, device=dev))
    return {"success": success, "steps": steps, "total_reward": total}


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
