"""Mixed-object batches (BASELINE config 5) in ONE simulator context and ONE stepping launch.

The reference swaps the object by loading a different MJCF per episode (kinova_gripper_env.py:986-1005, Latin-square
queue ENV:895-964).  Here a context holds the blobs of all the objects (ks_load_models); every env carries an object id
that ks_reset_objects may change at any reset, together with its randomised object mass / object-hand friction (an
extension beyond the reference, SURVEY 8d config 5: mass ~ U[0.05, 0.15] kg, mu ~ U[0.5, 1.0]; `scenarios.
config5_env_params` draws them).  Inside the library the envs are grouped by object (every stepping workgroup stages one
object's hull tables in LDS); results do not depend on the grouping: an env's trajectory is bit-identical to the same env
in a single-object context (tests/test_gpu_parity.py::test_mixed_shape_batch_equals_per_shape_contexts).

`MultiShapeSim` keeps round 1's interface (env i holds `shapes[i * len(shapes) // N]` unless `object_id` says
otherwise) on top of that single context.
"""
from __future__ import annotations

import numpy as np
import torch

from .sim import KinovaSim


class MultiShapeSim(KinovaSim):
    def __init__(self, n_envs: int, shapes, device: int = 0, object_id=None, **sim_kwargs):
        self.shapes = list(shapes)
        k = len(self.shapes)
        super().__init__(n_envs, self.shapes, device=device, **sim_kwargs)
        if object_id is None:
            base, extra = divmod(n_envs, k)
            counts = [base + (1 if i < extra else 0) for i in range(k)]
            object_id = np.repeat(np.arange(k), counts)
        self.shape_of_env = torch.as_tensor(np.asarray(object_id), dtype=torch.int32, device=self.device)
        self._mass_friction = None

    def reset(self, qpos0: torch.Tensor, hand_quat: torch.Tensor, env_ids=None, object_id=None, mass_friction=None):
        """All envs (or env_ids): object ids default to the env's current shape; mass_friction [2, n] defaults to what
        set_env_params stored (else every object's compiled values)."""
        if object_id is None:
            object_id = self.shape_of_env if env_ids is None else self.shape_of_env[torch.as_tensor(env_ids).long()]
        else:
            oid = torch.as_tensor(object_id, dtype=torch.int32, device=self.device)
            if env_ids is None:
                self.shape_of_env = oid.clone()
            else:
                self.shape_of_env[torch.as_tensor(env_ids).long()] = oid
        if mass_friction is None and self._mass_friction is not None:
            mass_friction = self._mass_friction if env_ids is None else self._mass_friction[:, torch.as_tensor(env_ids).long()]
        return super().reset(qpos0, hand_quat, env_ids, object_id=object_id, mass_friction=mass_friction)

    def set_env_params(self, obj_mass=None, obj_mu=None):
        """per-env object mass [N] and object-hand friction [N]: applied now and kept for later resets"""
        super().set_env_params(obj_mass, obj_mu)
        if obj_mass is not None and obj_mu is not None:
            self._mass_friction = torch.stack([torch.as_tensor(obj_mass), torch.as_tensor(obj_mu)]).to(self.device, self.dtype)
