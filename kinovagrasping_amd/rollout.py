"""Batched rollout + learner iteration (L5): the counterpart of update_policy in
gym-kinova-gripper/main_DDPGfD.py:333-537 for N envs stepping in lock step on one GPU.

Per env and episode, as in the reference:
  * after >= 6 steps, check_grasp(prev_obs[9:17], obs[9:17]) (expert_data.py:559-593) latches
    ready_for_lift (main_DDPGfD.py:426-439; `check_for_lift=False` after the first lift step);
  * not ready: a = clip(pi(s) + N(0, max_action*expl_noise), 0, max_action), transition stored
    (main_DDPGfD.py:443-451);
  * ready: scripted lift action [0.6, 0.5, 0.5, 0.5], NOT stored; when the episode ends during the lift
    the last stored transition's reward / not_done are overwritten with the outcome
    (main_DDPGfD.py:275-290, utils.py:309-343);
  * episodes with len - n <= 1 are dropped (main_DDPGfD.py:469-471).
Differences forced by batching (documented in DESIGN.md): all envs advance together, finished envs
are auto-reset inside ks_step, and the learner is stepped every env-step instead of 100 updates per
episode.  Everything here is torch on the GPU; the only native compute is libkinova_sim + the GEMMs.
"""
from __future__ import annotations

import torch

LIFT_ACTION = (0.6, 0.5, 0.5, 0.5)     # wrist_lift_velocity, finger_lift_velocity (main_DDPGfD.py:945-947)
SKIP_NUM_TS = 6                         # main_DDPGfD.py:418


def check_grasp(f_dist_old: torch.Tensor, f_dist_new: torch.Tensor) -> torch.Tensor:
    """Batched expert_data.check_grasp: inputs [N, 8] = obs[:, 9:17]; True where the summed |dx| of the
    three distal fingertips per substep is below 2e-4."""
    d = (f_dist_old[:, (0, 3, 6)] - f_dist_new[:, (0, 3, 6)]).abs() / 15.0
    return d.sum(1) < 0.0002


class RolloutEngine:
    def __init__(self, sim, policy, replay=None, expl_noise=0.1, max_action=0.8, generator=None):
        self.sim, self.policy, self.replay = sim, policy, replay
        self.n = sim.n_envs
        dev = sim.device
        self.sigma = max_action * expl_noise
        self.max_action = max_action
        self.gen = generator
        self.lift_action = torch.tensor(LIFT_ACTION, device=dev).expand(self.n, 4)
        self.t = torch.zeros(self.n, dtype=torch.long, device=dev)           # steps taken in the episode
        self.ready = torch.zeros(self.n, dtype=torch.bool, device=dev)       # ready_for_lift (latched)
        self.prev_obs = None
        self.obs = None
        self.episodes_done = 0
        self.lift_success = 0

    def start(self, obs):
        """obs [N, 82]: observations returned by the reset"""
        self.obs = obs.clone()
        self.prev_obs = None
        self.t.zero_()
        self.ready.zero_()

    @torch.no_grad()
    def act(self):
        timestep = self.t + 1                                              # main_DDPGfD.py:425
        if self.prev_obs is not None:
            chk = check_grasp(self.prev_obs[:, 9:17], self.obs[:, 9:17]) & (timestep >= SKIP_NUM_TS) & self.has_prev
            self.ready |= chk
        a = self.policy.actor(self.obs)
        noise = torch.randn(a.shape, device=a.device, generator=self.gen) * self.sigma
        a = (a + noise).clamp_(0.0, self.max_action)
        return torch.where(self.ready.unsqueeze(1), self.lift_action, a)

    @torch.no_grad()
    def step(self, after_act=None, after_launch=None, before_store=None):
        """One env-step for every env.  Returns (reward, done) of the step.  Hooks for running the learner
        beside the sim kernel on a second stream: `after_act()` right after the actor forward has been
        enqueued (record an event there), `after_launch()` right after the sim kernels have been enqueued
        (enqueue the update there), `before_store()` before the replay is written."""
        if self.prev_obs is None:
            self.has_prev = torch.zeros(self.n, dtype=torch.bool, device=self.obs.device)
            self.prev_obs = self.obs.clone()
        action = self.act()
        if after_act is not None:
            after_act()
        lifting = self.ready.clone()
        state = self.obs
        obs, reward, done, info = self.sim.step(action.t().contiguous())
        if after_launch is not None:
            after_launch()
        if before_store is not None:
            before_store()
        done_b = done != 0
        # the transition's next_state is the terminal observation for envs that just finished
        next_state = torch.where(done_b.unsqueeze(1), self.sim.final_obs, obs) if self.sim.cfg.auto_reset else obs
        if self.replay is not None:
            self.replay.add(state, action, next_state, reward, done_b, store_mask=~lifting)
            ended_lifting = done_b & lifting
            if ended_lifting.any():
                self.replay.replace_last(ended_lifting, reward)
            if done_b.any():
                self.replay.end_episodes(done_b)
        # bookkeeping for the next step (auto-reset envs start a new episode)
        self.prev_obs = torch.where(done_b.unsqueeze(1), obs, state)
        self.has_prev = ~done_b
        self.obs = obs.clone()
        self.t = torch.where(done_b, torch.zeros_like(self.t), self.t + 1)
        self.ready &= ~done_b
        return reward, done_b
