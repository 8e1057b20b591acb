"""TEST INFRASTRUCTURE: a handful of fp64 oracle envs behind the calling convention of kinovagrasping_amd.sim.KinovaSim
(reset / step with field-major torch tensors), so that the host-side loops of the package (demonstrators, evaluate) can be
run against the CPU oracle in `-m "not gpu"` tests exactly as they run against the HIP kernels in `-m gpu` tests."""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from kinovagrasping_amd import scenarios
from kinovagrasping_amd.sim import SOLVER_ITERATIONS
from oracle import ko_py as ko


class OracleVecSim:
    def __init__(self, n_envs: int, model: str = "CubeS", solver_iterations: int = SOLVER_ITERATIONS, horizon: int = 30, rays: bool = True,
                 narrow_phase: int = 0):
        self.n_envs, self.device = n_envs, torch.device("cpu")
        self.cfg = SimpleNamespace(auto_reset=0, horizon=horizon)
        self.model = ko.OracleModel(scenarios.model_blob(model))
        self.iters, self.rays, self.narrow_phase = solver_iterations, rays, narrow_phase
        self.sims = []
        self.final_obs = torch.zeros(n_envs, 82, dtype=torch.float64)
        self.t = 0

    def reset(self, qpos0, hand_quat):
        q, h = np.asarray(qpos0, dtype=np.float64), np.asarray(hand_quat, dtype=np.float64)
        self.sims = [ko.OracleSim(self.model, h[:, i].copy(), solver_iterations=self.iters) for i in range(self.n_envs)]
        for s in self.sims:
            s.s.rays_enabled = int(self.rays)
            s.s.narrow_phase = self.narrow_phase
        self.t = 0
        return torch.from_numpy(np.stack([s.env_reset(q[:, i].copy()) for i, s in enumerate(self.sims)]))

    def step(self, action):
        a = np.asarray(action, dtype=np.float64)
        obs, rew, done, info = np.zeros((self.n_envs, 82)), np.zeros(self.n_envs), np.zeros(self.n_envs, dtype=np.uint8), np.zeros((3, self.n_envs))
        self.t += 1
        for i, s in enumerate(self.sims):
            o, r, d, inf = s.env_step(a[:, i].copy())
            obs[i], rew[i], info[:, i] = o, r, inf
            done[i] = int(d) | (2 if self.t >= self.cfg.horizon else 0)
        return torch.from_numpy(obs), torch.from_numpy(rew), torch.from_numpy(done), torch.from_numpy(info)

    def qpos(self):
        return np.stack([s.view("qpos").copy() for s in self.sims], 1)


def place_at_palm_xy(sim, x, y, shape: str = "CubeS", orientation: str = "normal"):
    """Start configurations [16, n] that put the object's centre at the palm-frame coordinates (x[i], y[i]) the reference
    records for a start (obj_local = Tfw . obj_world, expert_data.py:729-732): `sim` is anything with reset(qpos0, hand_quat)
    -> obs whose slots 21:23 are that palm-frame position (KinovaSim or OracleVecSim).  The map world (X, Y) -> palm
    (x, y) at fixed world Z is affine, so one finite-difference Jacobian and one correction are exact."""
    n = sim.n_envs
    x, y = np.broadcast_to(np.asarray(x, dtype=np.float64), (n,)), np.broadcast_to(np.asarray(y, dtype=np.float64), (n,))
    z0 = float(scenarios.start_coord_table(shape, orientation)[0][2])
    hq = np.repeat(scenarios.hand_quat_for(orientation)[:, None], n, 1)

    def local_xy(wx, wy):
        q = np.zeros((16, n))
        q[9], q[10], q[11], q[12] = wx, wy, z0, 1.0
        o = sim.reset(torch.as_tensor(q), torch.as_tensor(hq))
        return np.asarray(o.double().cpu())[:, 21:23].T.copy(), q

    w = np.zeros((2, n))
    h = 1e-2
    p0, _ = local_xy(w[0], w[1])
    px, _ = local_xy(w[0] + h, w[1])
    py, _ = local_xy(w[0], w[1] + h)
    J = np.stack([(px - p0) / h, (py - p0) / h], -1)               # [2 (local), n, 2 (world)]
    for _ in range(2):
        p, _ = local_xy(w[0], w[1])
        r = np.stack([x, y]) - p
        for i in range(n):
            w[:, i] += np.linalg.solve(J[:, i, :], r[:, i])
    p, q = local_xy(w[0], w[1])
    return q, hq, p
