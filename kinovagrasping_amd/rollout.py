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
    d = (f_dist_old[:, 0:7:3] - f_dist_new[:, 0:7:3]).abs() / 15.0     # x of the three fingertips (columns 0, 3, 6)
    return d.sum(1) < 0.0002


class RolloutEngine:
    """State lives in fixed tensors that are updated in place and every op has a fixed shape, so `pre()` (action
    selection) and `post()` (replay + bookkeeping) can each be captured in a HIP graph (pipeline.GraphedTrainer);
    `step()` is the plain eager sequence pre -> sim -> post."""

    def __init__(self, sim, policy, replay=None, expl_noise=0.1, max_action=0.8, generator=None, device_noise=None):
        """device_noise: draw the exploration noise inside the fused actor kernel (counter-based Philox keyed by
        torch.initial_seed(), a device step counter and the env index) instead of with torch.randn.  Default: on when no
        torch generator is given and the actor is a 3-layer MLP the fused kernel supports."""
        self.sim, self.policy, self.replay = sim, policy, replay
        self.n = sim.n_envs
        dev = sim.device
        self.sigma = max_action * expl_noise
        self.max_action = max_action
        self.gen = generator
        self.lift_action = torch.tensor(LIFT_ACTION, device=dev).expand(self.n, 4)
        self.t = torch.zeros(self.n, dtype=torch.long, device=dev)           # steps taken in the episode
        self.ready = torch.zeros(self.n, dtype=torch.bool, device=dev)       # ready_for_lift (latched)
        self.lifting = torch.zeros(self.n, dtype=torch.bool, device=dev)     # ready at the time of the action
        self.has_prev = torch.zeros(self.n, dtype=torch.bool, device=dev)
        self.obs = torch.zeros(self.n, 82, device=dev)
        self.prev_obs = torch.zeros(self.n, 82, device=dev)
        self.action = torch.zeros(self.n, 4, device=dev)
        self.action_t = torch.zeros(4, self.n, device=dev)                   # field-major copy for ks_step
        self.reward_out = torch.zeros(self.n, device=dev)
        self.done_out = torch.zeros(self.n, dtype=torch.bool, device=dev)
        self.episodes_done = 0
        self.lift_success = 0
        # on a GPU the per-env rules run as two kernels of libkinova_sim.so (include/kinova_rollout.h); the torch
        # code in pre() / post() is the same arithmetic and the checker of those kernels
        self.native = dev.type == "cuda" and (replay is None or getattr(replay, "native", False))
        if self.native:
            from . import sim as _sim
            self._lib, self._ptr = _sim.load_library(), _sim._ptr
            self._keep = torch.zeros(self.n, dtype=torch.bool, device=dev)
        self.device_noise = (generator is None) if device_noise is None else bool(device_noise)
        self.noise_seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self.rng_state = torch.zeros(2, dtype=torch.long, device=dev)        # [launch counter, scratch] of kr_actor_select

    def _stream(self):
        import ctypes
        return ctypes.c_void_p(torch.cuda.current_stream(self.sim.device).cuda_stream)

    def start(self, obs):
        """obs [N, 82]: observations returned by the reset"""
        self.obs.copy_(obs)
        self.prev_obs.copy_(obs)
        self.has_prev.zero_()
        self.t.zero_()
        self.ready.zero_()

    @torch.no_grad()
    def pre(self):
        """Action selection for every env -> self.action [N,4] / self.action_t [4,N]."""
        if self.native:
            layers = self._fused_actor_layers()
            if layers is not None:
                # actor forward + noise + selection rule: one launch (kr_actor_select, csrc/ks_mlp.hip)
                (w1, b1), (w2, b2), (w3, b3) = layers
                P = self._ptr
                noise = None if self.device_noise else torch.randn(self.n, 4, device=self.obs.device, generator=self.gen)
                rc = self._lib.kr_actor_select(self.n, w1.shape[0], w2.shape[0], P(self.obs), P(self.prev_obs), P(self.has_prev), P(self.t),
                                               P(self.ready), P(w1), P(b1), P(w2), P(b2), P(w3), P(b3), P(noise), self.noise_seed,
                                               P(self.rng_state) if noise is None else None, self.sigma, self.max_action, SKIP_NUM_TS, None,
                                               P(self.action), P(self.action_t), P(self.lifting), self._stream())
                if rc != 0:
                    raise RuntimeError(f"kr_actor_select failed ({rc})")
                return
            a = self.policy.actor(self.obs).contiguous()
            noise = torch.randn(a.shape, device=a.device, generator=self.gen)
            P = self._ptr
            rc = self._lib.kr_select_action(self.n, P(self.obs), P(self.prev_obs), P(self.has_prev), P(self.t), P(self.ready), P(a), P(noise),
                                            self.sigma, self.max_action, SKIP_NUM_TS, P(self.action), P(self.action_t), P(self.lifting),
                                            self._stream())
            if rc != 0:
                raise RuntimeError(f"kr_select_action failed ({rc})")
            return
        timestep = self.t + 1                                              # main_DDPGfD.py:425
        chk = check_grasp(self.prev_obs[:, 9:17], self.obs[:, 9:17]) & (timestep >= SKIP_NUM_TS) & self.has_prev
        self.ready |= chk
        a = self.policy.actor(self.obs)
        noise = torch.randn(a.shape, device=a.device, generator=self.gen) * self.sigma
        a = (a + noise).clamp_(0.0, self.max_action)
        self.action.copy_(torch.where(self.ready.unsqueeze(1), self.lift_action, a))
        self.action_t.copy_(self.action.t())
        self.lifting.copy_(self.ready)

    def _fused_actor_layers(self):
        """the actor's (weight, bias) x 3 when it is the reference's 3-layer MLP at a width the fused kernel supports
        (max_action * sigmoid output), else None"""
        actor = self.policy.actor
        if not all(hasattr(actor, k) for k in ("l1", "l2", "l3")):
            return None
        from . import mlp
        layers = mlp.layers_of(actor)
        ok = mlp.supported(layers, self.obs.shape[1]) and layers[2][0].shape[0] == 4 and float(getattr(actor, "max_action", -1.0)) == float(self.max_action)
        return layers if ok else None

    def act(self):
        self.pre()
        return self.action

    @torch.no_grad()
    def commit(self):
        """Second half of post(commit=False): move the episodes that finished in the last stored step from the open-episode
        buffers into the replay ring (kr_rank_episodes / kr_commit_episodes / kr_advance_ring).  Must run before the next
        post(); pipeline.GraphedTrainer runs it on the learner's stream at the start of the next env-step."""
        if self.native and self.replay is not None:
            self.replay.commit_native(self._keep, self.done_out)

    @torch.no_grad()
    def post(self, commit=True):
        """Consume the sim's output buffers: replay writes and per-env bookkeeping for the next step.  commit=False
        (kernel path only) leaves the ring update to commit()."""
        sim = self.sim
        if self.native:
            rp, P = self.replay, self._ptr
            N = lambda *a: [None] * len(a) if rp is None else list(a)
            cur = N("cur_state", "cur_next", "cur_action", "cur_reward", "cur_not_done", "cur_len")
            cur = [None if c is None else P(getattr(rp, c)) for c in cur]
            rc = self._lib.kr_store_transition(self.n, rp.horizon if rp else 1, rp.n_steps if rp else 0, int(sim.cfg.auto_reset), int(rp is not None),
                                               P(sim.obs), P(sim.final_obs), P(sim.reward), P(sim.done), P(self.obs), P(self.prev_obs),
                                               P(self.has_prev), P(self.t), P(self.ready), P(self.lifting), P(self.action), *cur,
                                               P(self.reward_out), P(self.done_out), P(self._keep), self._stream())
            if rc != 0:
                raise RuntimeError(f"kr_store_transition failed ({rc})")
            if rp is not None and commit:
                rp.commit_native(self._keep, self.done_out)
            return
        obs, reward = sim.obs, sim.reward
        done_b = sim.done != 0
        state, lifting = self.obs, self.lifting
        # the transition's next_state is the terminal observation for envs that just finished
        next_state = torch.where(done_b.unsqueeze(1), sim.final_obs, obs) if sim.cfg.auto_reset else obs
        if self.replay is not None:
            self.replay.add(state, self.action, next_state, reward, done_b, store_mask=~lifting)
            self.replay.replace_last(done_b & lifting, reward)
            self.replay.end_episodes(done_b)
        # bookkeeping for the next step (auto-reset envs start a new episode)
        self.prev_obs.copy_(torch.where(done_b.unsqueeze(1), obs, state))
        self.has_prev.copy_(~done_b)
        self.obs.copy_(obs)
        self.t.copy_(torch.where(done_b, torch.zeros_like(self.t), self.t + 1))
        self.ready &= ~done_b
        self.reward_out.copy_(reward)
        self.done_out.copy_(done_b)

    def step(self, after_act=None, after_launch=None, before_store=None):
        """One env-step for every env.  Returns (reward, done) of the step.  Hooks for running the learner
        beside the sim kernel on a second stream: `after_act()` right after the actor forward has been
        enqueued (record an event there), `after_launch()` right after the sim kernels have been enqueued
        (enqueue the update there), `before_store()` before the replay is written."""
        self.pre()
        if after_act is not None:
            after_act()
        self.sim.step(self.action_t)
        if after_launch is not None:
            after_launch()
        if before_store is not None:
            before_store()
        self.post()
        return self.reward_out, self.done_out
