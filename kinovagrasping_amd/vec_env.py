"""Host-side mirror of the reference env interface for N batched envs on one MI355X.

Keeps the names, argument meaning and return layout of KinovaGripper_Env
(gym-kinova-gripper/gym_kinova_gripper/envs/kinova_gripper_env.py): reset(...) -> observations,
step(action) -> (obs, reward, done, info) with info keys finger_reward / grasp_reward / lift_reward
(ENV:685), action_space Box(+-0.8, (4,)) (ENV:128), _max_episode_steps (main_DDPGfD.py:384),
get_obj_coords / get_orientation (main_DDPGfD.py:167,253).  Batched: every quantity gains a leading
env dimension and lives on the GPU as a torch tensor; the compute is libkinova_sim.so.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from . import scenarios
from .sim import KinovaSim, NOBS


class KinovaGripperVecEnv:
    metadata = {"render.modes": []}

    def __init__(self, n_envs: int, shape: str = "CubeS", device: int = 0, frame_skip: int = 15, max_episode_steps: int = 30,
                 auto_reset: bool = True, solver_iterations: int = 6, seed: int = 0):
        self.n_envs = n_envs
        self.random_shape = shape
        self.frame_skip = frame_skip
        self._max_episode_steps = max_episode_steps
        self.action_space = SimpleNamespace(low=np.full(4, -0.8, np.float32), high=np.full(4, 0.8, np.float32), shape=(4,), dtype=np.float32)
        self.observation_dim = NOBS
        self.sim = KinovaSim(n_envs, shape, device=device, frame_skip=frame_skip, horizon=max_episode_steps,
                             solver_iterations=solver_iterations, auto_reset=auto_reset, obs_env_major=True)
        self.np_random = np.random.RandomState(seed)
        self.orientation = ["normal"] * n_envs
        self.obj_coords = np.zeros((n_envs, 3))
        self.with_grasp_reward = False

    # -- reference-compatible accessors -----------------------------------------------------------
    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def get_obj_coords(self):
        return self.obj_coords

    def get_orientation(self):
        return self.orientation

    def get_random_shape(self):
        return self.random_shape

    def set_with_grasp_reward(self, with_grasp):
        if with_grasp:
            raise NotImplementedError("the grasp-classifier reward needs gc_model.pkl, which the reference does not ship "
                                      "(.MISSING_LARGE_BLOBS:1); default False at main_DDPGfD.py:906")
        self.with_grasp_reward = False

    # -- reset ------------------------------------------------------------------------------------
    def select_orienation(self, hand_orientation: str):
        """ENV:1180-1222 (scenarios.select_orientation, pinned by tests/golden/schedule.npz); a fixed class name
        ('normal' / 'rotated' / 'top') is taken as is."""
        if hand_orientation in scenarios.ORIENTATION_EULER:
            return hand_orientation
        return scenarios.select_orientation(self.random_shape, hand_orientation, self.np_random)

    def reset(self, shape_keys=None, hand_orientation="normal", with_grasp=False, env_name="env", mode="train", start_pos=None,
              obj_params=None, qpos=None, obj_coord_region=None, with_noise=False, env_ids=None):
        """Reset all envs (or `env_ids`).  start_pos: optional [n,3] object positions; otherwise rows are
        sampled from the no_noise start-coordinate table of the shape (ENV:1008-1054, SURVEY note N5).
        Slide offsets are zero for every orientation: determine_hand_location (ENV:1286-1307) multiplies by the
        env's Tfw *before* it is first computed, and the training loop builds a fresh env (Tfw = zeros, ENV:114)
        for every episode (main_DDPGfD.py:381), so the offsets it actually uses are 0."""
        self.set_with_grasp_reward(with_grasp)
        ids = np.arange(self.n_envs) if env_ids is None else np.asarray(env_ids)
        n = len(ids)
        q = np.zeros((16, n))
        hq = np.zeros((4, n))
        q[12] = 1.0
        for k, e in enumerate(ids):
            o = self.select_orienation(hand_orientation)
            self.orientation[e] = o
            hq[:, k] = scenarios.hand_quat_for(o)
            if start_pos is not None:
                q[9:12, k] = np.asarray(start_pos)[k][:3]
            else:
                tab = scenarios.start_coord_table(self.random_shape, o)
                q[9:12, k] = tab[self.np_random.randint(0, len(tab))]
            self.obj_coords[e] = q[9:12, k]
        t_ids = None if env_ids is None else torch.as_tensor(ids, dtype=torch.int32)
        obs = self.sim.reset(torch.as_tensor(q), torch.as_tensor(hq), t_ids)
        return obs

    # -- step -------------------------------------------------------------------------------------
    def step(self, action, graspnetwork=False):
        """action: [N,4] (wrist, finger1..3) torch tensor or array.  Returns (obs [N,82], reward [N],
        done [N] bool, info dict of [N] tensors)."""
        a = torch.as_tensor(action, dtype=torch.float32, device=self.sim.device)
        if a.shape != (self.n_envs, 4):
            raise ValueError(f"action must be [{self.n_envs}, 4], got {tuple(a.shape)}")
        obs, reward, done, info = self.sim.step(a.t().contiguous())
        infod = {"finger_reward": info[0], "grasp_reward": info[1], "lift_reward": info[2],
                 "TimeLimit.truncated": (done & 2).bool() & ~(done & 1).bool(), "final_obs": self.sim.final_obs}
        return obs, reward, done.bool(), infod

    def close(self):
        self.sim.close()
