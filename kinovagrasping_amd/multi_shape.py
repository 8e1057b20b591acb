"""Mixed-object batches (BASELINE config 5): one KinovaSim context per object shape, stepped concurrently
on separate HIP streams, presented as one batch.

The reference swaps the object by loading a different MJCF per episode (kinova_gripper_env.py:986-1005,
Latin-square queue ENV:895-964); here env i keeps the shape `shapes[i * len(shapes) // N]` for its whole
life, which is the in-memory replacement SURVEY.md section 2 row 17 describes.  Per-env object mass /
object-hand friction (an extension beyond the reference, SURVEY 8d config 5: mass ~ U[0.05, 0.15] kg,
mu ~ U[0.5, 1.0]) go through `set_env_params` (ks_set_env_params); `scenarios.config5_env_params` draws them.
"""
from __future__ import annotations

import torch

from .sim import KinovaSim, NOBS


class MultiShapeSim:
    def __init__(self, n_envs: int, shapes, device: int = 0, **sim_kwargs):
        self.shapes = list(shapes)
        k = len(self.shapes)
        base, extra = divmod(n_envs, k)
        self.counts = [base + (1 if i < extra else 0) for i in range(k)]
        self.offsets = [sum(self.counts[:i]) for i in range(k + 1)]
        self.n_envs = n_envs
        self.device = torch.device("cuda", device)
        self.sims = [KinovaSim(c, s, device=device, **sim_kwargs) for c, s in zip(self.counts, self.shapes)]
        self.streams = [torch.cuda.Stream(self.device) for _ in self.sims]
        self.cfg = self.sims[0].cfg
        self.shape_of_env = torch.repeat_interleave(torch.arange(k), torch.tensor(self.counts)).to(self.device)

    def _fan_out(self, fn):
        main = torch.cuda.current_stream(self.device)
        outs = []
        for i, (sim, st) in enumerate(zip(self.sims, self.streams)):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                outs.append(fn(i, sim))
        for st in self.streams:
            main.wait_stream(st)
        return outs

    def reset(self, qpos0: torch.Tensor, hand_quat: torch.Tensor):
        qpos0, hand_quat = qpos0.to(self.device), hand_quat.to(self.device)
        o = self.offsets
        outs = self._fan_out(lambda i, sim: sim.reset(qpos0[:, o[i]:o[i + 1]], hand_quat[:, o[i]:o[i + 1]]))
        return torch.cat(outs, 0)

    def step(self, action: torch.Tensor):
        """action [4, N] -> (obs [N,82], reward [N], done [N] uint8, info [3,N])"""
        action = action.to(self.device)
        o = self.offsets
        outs = self._fan_out(lambda i, sim: sim.step(action[:, o[i]:o[i + 1]]))
        self.final_obs = torch.cat([s.final_obs for s in self.sims], 0)
        # the same output attributes a single KinovaSim keeps (rollout.RolloutEngine reads them)
        self.obs, self.reward, self.done = (torch.cat([x[k] for x in outs], 0) for k in range(3))
        return self.obs, self.reward, self.done, torch.cat([x[3] for x in outs], 1)

    def set_env_params(self, obj_mass=None, obj_mu=None):
        """per-env object mass [N] and object-hand friction [N] (None = leave as is)"""
        o = self.offsets
        for i, sim in enumerate(self.sims):
            sim.set_env_params(None if obj_mass is None else obj_mass[o[i]:o[i + 1]], None if obj_mu is None else obj_mu[o[i]:o[i + 1]])

    def get_state(self):
        sts = [s.get_state() for s in self.sims]
        return {k: torch.cat([st[k] for st in sts], -1) for k in sts[0]}

    def close(self):
        for s in self.sims:
            s.close()
