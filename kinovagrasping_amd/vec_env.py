"""Host-side mirror of the reference env interface for N batched envs on one MI355X.

Keeps the names, argument meaning and return layout of KinovaGripper_Env
(gym-kinova-gripper/gym_kinova_gripper/envs/kinova_gripper_env.py): reset(...) -> observations,
step(action) -> (obs, reward, done, info) with info keys finger_reward / grasp_reward / lift_reward
(ENV:685), action_space Box(+-0.8, (4,)) (ENV:128), _max_episode_steps (main_DDPGfD.py:384),
get_obj_coords / get_orientation (main_DDPGfD.py:167,253), Tfw (main_DDPGfD.py:170,406), get_orientation_idx
(main_DDPGfD.py:411), get_coords_filename (main_DDPGfD.py:161), Generate_Latin_Square / check_obj_file_empty
(main_DDPGfD.py:387-388).  Batched: every quantity gains a leading env dimension and lives on the GPU as a torch tensor;
the compute is libkinova_sim.so (libkinova_sim_mg.so when a multi-geom object - Bottle / TBottle / Bowl / RBowl - is among the shapes).  With a LIST of shapes the env holds all those objects in one simulator context and
every reset picks each env's object the way the reference's reset() does per episode (select_object, ENV:986-1005).
"""
from __future__ import annotations

import csv
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import scenarios
from .model_compiler import read_blob
from .sim import NOBS, SOLVER_ITERATIONS, KinovaSim

COORDS_DIR = "gym_kinova_gripper/envs/kinova_description/obj_hand_coords/"     # ENV:1245


class KinovaGripperVecEnv:
    metadata = {"render.modes": []}

    def __init__(self, n_envs: int, shape="CubeS", device: int = 0, frame_skip: int = 15, max_episode_steps: int = 30,
                 auto_reset: bool = True, solver_iterations: int = SOLVER_ITERATIONS, seed: int = 0, hand_offsets: str = "fresh-env",
                 host_only: bool = False):
        """shape: one object name, or a list of them (mixed-object batches, BASELINE config 5).
        hand_offsets: where the 'rotated' / 'top' hands start (determine_hand_location, ENV:1286-1307) - "fresh-env" (default: what the
        reference's drivers do): zero, because main_DDPGfD.py:381 / expert_data.py recreate the env for every episode and a fresh env's
        Tfw is zero; "pose": with the pose's own palm rotation - what determine_hand_location is written to do, the hand hovers over
        the object - (scenarios.hand_slide_offsets; BASELINE config 5 and bench.py ask for "pose" explicitly).
        host_only: no device, no simulator - reset() then RETURNS the start states it drew ({"qpos": [16, n], "hand_quat": [4, n]}) instead
        of observations: the reset's sampling is host code and is pinned against the reference's on CPU (tests/test_reset_golden.py)."""
        if hand_offsets not in ("fresh-env", "pose"):
            raise ValueError('KinovaGripperVecEnv: hand_offsets is "fresh-env" (the reference drivers\' zero offsets) or "pose"')
        self.hand_offsets = hand_offsets
        self.n_envs = n_envs
        self.shapes = [shape] if isinstance(shape, str) else list(shape)
        self.random_shape = self.shapes[0] if isinstance(shape, str) else [self.shapes[0]] * n_envs
        self.shape_id = np.zeros(n_envs, dtype=np.int32)
        self.frame_skip = frame_skip
        self._max_episode_steps = max_episode_steps
        self.action_space = SimpleNamespace(low=np.full(4, -0.8, np.float32), high=np.full(4, 0.8, np.float32), shape=(4,), dtype=np.float32)
        self.observation_dim = NOBS
        self.sim = None if host_only else KinovaSim(n_envs, shape if isinstance(shape, str) else self.shapes, device=device, frame_skip=frame_skip,
                                                    horizon=max_episode_steps, solver_iterations=solver_iterations, auto_reset=auto_reset, obs_env_major=True)
        self.np_random = np.random.RandomState(seed)
        self.orientation = ["normal"] * n_envs
        self.obj_coords = np.zeros((n_envs, 3))
        self.orientation_idx = np.zeros(n_envs, dtype=np.int64)
        self.hand_quat = np.repeat(scenarios.hand_quat_for("normal")[:, None], n_envs, 1)
        self.hand_euler = np.repeat(np.asarray(scenarios.ORIENTATION_EULER["normal"])[None], n_envs, 0)   # as patched into the XML (truncated)
        self.with_noise = "tables"
        self.obj_keys = []                       # Latin-square object queue (Generate_Latin_Square); reset() pops from its end
        self.with_grasp_reward = False
        M = read_blob(scenarios.model_blob(self.shapes[0]))        # hand constants for Tfw (the same in every object's blob)
        self._l7_pos, self._slide_axis = M["body_pos"][2], M["slide_axis"]
        self._palm_pos, self._palm_quat = M["geom_pos"][1], M["geom_quat"][1]
        self._obj_geom_pos = {}

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

    def get_orientation_idx(self):
        """row of the start-coordinate table every env's episode started from (ENV:1044, main_DDPGfD.py:411)"""
        return self.orientation_idx

    def get_coords_filename(self):
        """the coordinate file of every env's (orientation, shape), the path the reference builds at ENV:1245"""
        names = self.random_shape if isinstance(self.random_shape, list) else [self.random_shape] * self.n_envs
        noise_dir = "with_noise" if self.with_noise == "tables" else "no_noise"
        return [COORDS_DIR + noise_dir + "/" + getattr(self, "mode", "train") + "_coords/" + o + "/" + sh + ".txt" for o, sh in zip(self.orientation, names)]

    @property
    def Tfw(self):
        """[N, 4, 4] world -> palm-local transforms (ENV:274-288: T = (R_palm C)^T, Tfw = [T, -T wrist]) of the envs'
        CURRENT joint state (the reference refreshes its copy inside _get_obs, i.e. at the pose the last observation
        saw; right after a reset - where the drivers read it, main_DDPGfD.py:170,406 - the two are the same)."""
        from .model_compiler import quat_to_mat
        qpos = self.sim.get_state()["qpos"].double().cpu().numpy()
        C_ = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=np.float64)
        Rg = quat_to_mat(self._palm_quat)
        out = np.zeros((self.n_envs, 4, 4))
        for e in range(self.n_envs):
            R7 = quat_to_mat(self.hand_quat[:, e] / np.linalg.norm(self.hand_quat[:, e]))
            p7 = self._l7_pos + R7 @ (self._slide_axis.T @ qpos[0:3, e])
            Rp, pp = R7 @ Rg, p7 + R7 @ self._palm_pos
            T = (Rp @ C_).T
            wrist = pp + T.T @ np.array([-0.009, 0.048, 0.0])
            out[e, :3, :3], out[e, :3, 3], out[e, 3, 3] = T, -T @ wrist, 1.0
        return out

    def _get_obj_pose(self):
        """[N, 3] world position of the geom named `object` at the envs' current state (ENV:564-566; the demonstration drivers read it,
        expert_data.py:207-208, 249)"""
        from .model_compiler import quat_to_mat
        qpos = self.sim.get_state()["qpos"].double().cpu().numpy()
        names = self.random_shape if isinstance(self.random_shape, list) else [self.random_shape] * self.n_envs
        out = np.zeros((self.n_envs, 3))
        for e in range(self.n_envs):
            if names[e] not in self._obj_geom_pos:
                self._obj_geom_pos[names[e]] = read_blob(scenarios.model_blob(names[e]))["geom_pos"][8].copy()
            qn = qpos[12:16, e] / np.linalg.norm(qpos[12:16, e])
            out[e] = qpos[9:12, e] + quat_to_mat(qn) @ self._obj_geom_pos[names[e]]
        return out

    def _get_dot_product(self, obj_state=None):
        """[N] ENV:591-609: 20th power of the dot product of the unit vectors |object - link_7| and |origin - link_7| in the world x-y plane (absolute components, as
        the reference takes them); obj_state: [N, 3] object positions, default the current ones"""
        from .model_compiler import quat_to_mat
        obj = self._get_obj_pose() if obj_state is None else np.asarray(obj_state, dtype=np.float64).reshape(self.n_envs, -1)
        qpos = self.sim.get_state()["qpos"].double().cpu().numpy()
        out = np.zeros(self.n_envs)
        for e in range(self.n_envs):
            R7 = quat_to_mat(self.hand_quat[:, e] / np.linalg.norm(self.hand_quat[:, e]))
            hand = self._l7_pos + R7 @ (self._slide_axis.T @ qpos[0:3, e])
            ov, cv = np.abs(obj[e, :2] - hand[:2]), np.abs(0.0 - hand[:2])
            out[e] = float((ov / np.linalg.norm(ov)) @ (cv / np.linalg.norm(cv))) ** 20            # "cuspy to get distinct reward" (ENV:608)
        return out

    def get_all_objects(self):
        """the reference's object table (ENV:150-208, main_DDPGfD.py:1269): shape key -> model; here the compiled asset of each of the 42 keys"""
        return {k: str(scenarios.ASSETS / f"{k}.ksm") for k in scenarios.SHAPES + scenarios.MEDIUM_SHAPES + scenarios.EXTRA_SHAPES + scenarios.MULTI_GEOM_SHAPES}

    # -- object schedule (ENV:884-1005) --------------------------------------------------------------
    def check_obj_file_empty(self, filename):
        """ENV:884-893: False when the file does not exist (sic), True when it exists and is empty"""
        if not os.path.exists(filename):
            return False
        with open(filename, "r") as f:
            return not f.read(1)

    def Generate_Latin_Square(self, max_elements, filename, shape_keys, test=False):
        """ENV:895-964: the cyclic-rotation object list (scenarios.latin_square_object_keys, pinned by
        tests/golden/schedule.npz), kept as the env's queue and written to `filename` in the reference's csv form (one
        key per row, one character per column - csv.writer.writerow on a string)."""
        self.obj_keys = self.obj_keys + scenarios.latin_square_object_keys(list(shape_keys), max_elements)
        with open(filename, "w", newline="") as out:
            w = csv.writer(out)
            for key in self.obj_keys:
                w.writerow(key)

    def get_obj_keys(self):
        return self.obj_keys

    def select_object(self, shape_keys):
        """one env's object for its next episode: popped from the END of the Latin-square queue when there is one
        (get_object, ENV:986-989), else drawn uniformly from shape_keys"""
        if self.obj_keys:
            return self.obj_keys.pop()
        return shape_keys[self.np_random.randint(0, len(shape_keys))]

    def _obj_size_last(self, shape):
        """_get_obj_size()[-1] of a shape (the observation's third size slot is twice that, ENV:529)"""
        return float(read_blob(scenarios.model_blob(shape))["obj_size_obs"][2]) / 2.0

    def set_with_grasp_reward(self, with_grasp):
        if with_grasp:
            raise NotImplementedError("the grasp-classifier reward needs gc_model.pkl, which the reference does not ship "
                                      "(.MISSING_LARGE_BLOBS:1); default False at main_DDPGfD.py:906")
        self.with_grasp_reward = False

    # -- reset ------------------------------------------------------------------------------------
    def select_orienation(self, hand_orientation: str, shape=None):
        """ENV:1180-1222 (scenarios.select_orientation, pinned by tests/golden/schedule.npz); a fixed class name
        ('normal' / 'rotated' / 'top') is taken as is."""
        if hand_orientation in scenarios.ORIENTATION_EULER:
            return hand_orientation
        return scenarios.select_orientation(shape if shape is not None else self.random_shape, hand_orientation, self.np_random)

    def reset(self, shape_keys=None, hand_orientation="normal", with_grasp=False, env_name="env", mode="train", start_pos=None,
              obj_params=None, qpos=None, obj_coord_region=None, with_noise=True, env_ids=None):
        """Reset all envs (or `env_ids`).  start_pos: optional per-env rows as the reference's test hook takes them (ENV:1347-1363): 3 values =
        object x, y, z; 2 values = object x, y at height _get_obj_size()[-1]; 9 values = the three slides, the three proximal
        joints, object x, y, z.  Otherwise rows are
        sampled from the no_noise start-coordinate table of the shape (ENV:1008-1054, SURVEY note N5).
        obj_coord_region (ENV:1032-1046): "left" / "center" / "target" / "right" = a random row among those whose x
        lies in the region - indexed, as the reference does, into the WHOLE file with the index it drew from the region's rows -; "origin" = (0, 0)
        at the height of the file's first row.
        obj_params: the reference's [shape, size] test hook (ENV:1171-1178: every reset env gets object shape + size, which must be among the
        env's objects); a [2, n] ARRAY here is the config-5
        extension (per-env object mass, object-hand friction).  qpos: [n, 16] full joint vectors written as they are (`set_sim_state`, ENV:349-353).
        With several objects loaded every reset env draws its
        object from `shape_keys` (default: all loaded) - Latin-square queue first, see select_object.
        Slide offsets of the 'rotated' / 'top' hands: see `hand_offsets` of the constructor.
        mode ("train" / "test", ENV:1241-1245): which of the reference's two sets of coordinate files the rows come from (<noise>/train_coords: 4499 rows per
        file, <noise>/test_coords: 499).
        with_noise=True (the default, as in the reference, ENV:1310; any non-string flag counts by its truth value): the reference's start states AS THEY ARE - object position AND hand
        Euler triple of a random row of the shape's with_noise coordinate file (ENV:1019-1021, 1254-1255; `scenarios.noisy_start_table`),
        the Euler triple through the reference's 5-character truncation (ENV:870-874); the tables' bias of -0.087 rad and the swap
        between the normal / top classes (SURVEY note N5; generator rotation_generation.py:20-25) are the reference's and are kept.
        Where the reference ships no such file the no-noise path applies.  ("tables" is accepted as an alias.)
        with_noise=False: the class's no-noise Euler constants (ENV:1267-1273) and a row of the no_noise table.
        with_noise="zero-mean": an EXTENSION (what SURVEY note N5 prescribes instead of the biased tables) - the no-noise constants plus
        ZERO-MEAN N(0, 0.087 rad) per axis drawn from the env's np_random, then the 5-character truncation; object coordinates from the
        no_noise tables."""
        from .model_compiler import euler_to_quat, truncated_euler
        if not isinstance(with_noise, str):                     # bool / np.bool_ / int flags forwarded by a driver (argparse, numpy): truth value
            with_noise = "tables" if bool(with_noise) else False
        if with_noise not in (False, "tables", "zero-mean"):
            raise ValueError('reset: with_noise is True (the reference\'s with_noise files; alias "tables"), False or "zero-mean" (extension)')
        if mode not in ("train", "test"):
            raise ValueError(f"reset: mode is 'train' or 'test' (the coordinate files' directory, ENV:1241-1245), not {mode!r}")
        self.with_noise, self.mode = with_noise, mode
        self.set_with_grasp_reward(with_grasp)
        ids = np.arange(self.n_envs) if env_ids is None else np.asarray(env_ids)
        n = len(ids)
        q = np.zeros((16, n))
        hq = np.zeros((4, n))
        q[12] = 1.0
        multi = isinstance(self.random_shape, list)
        keys = [k for k in (shape_keys or self.shapes)]
        forced = None
        if isinstance(obj_params, (list, tuple)):
            if len(obj_params) != 2 or not all(isinstance(x, str) for x in obj_params):
                raise ValueError("reset: obj_params is [shape, size] (e.g. ['Cube', 'S']) or a [2, n] array of mass / friction")
            forced = obj_params[0] + obj_params[1]
            if forced not in self.shapes:
                raise ValueError(f"reset: obj_params {obj_params} -> {forced}, not among the env's objects {self.shapes}")
        regions = {"left": (-.09, -.03), "center": (-.03, .03), "target": (-.01, .01), "right": (.03, .09)}
        if obj_coord_region is not None and obj_coord_region != "origin" and obj_coord_region not in regions:
            raise ValueError(f"reset: obj_coord_region {obj_coord_region!r}")
        if multi and any(k not in self.shapes for k in keys):
            raise ValueError(f"shape_keys {keys} not all among the env's objects {self.shapes}")
        for k, e in enumerate(ids):
            if multi:
                name = forced if forced is not None else self.select_object(keys)
                self.random_shape[e], self.shape_id[e] = name, self.shapes.index(name)
            shape = self.random_shape[e] if multi else self.random_shape
            o = self.select_orienation(hand_orientation, shape)
            self.orientation[e] = o
            def pick_row(tab):
                """row index as sample_initial_object_hand_pos draws it (ENV:1029-1048); -1 = the "origin" region"""
                if obj_coord_region == "origin":
                    return -1
                if obj_coord_region is not None:
                    lo, hi = regions[obj_coord_region]
                    return self.np_random.randint(0, int(((tab[:, 0] >= lo) & (tab[:, 0] <= hi)).sum()))     # (sic: an index INTO the region's rows,
                return self.np_random.randint(0, len(tab))                                                 #  used on the whole file)

            noisy = scenarios.noisy_start_table(shape, o, mode) if (with_noise == "tables" and start_pos is None) else None
            if noisy is not None:
                row = pick_row(noisy)
                eul = truncated_euler(noisy[row, 3:6]) if row >= 0 else np.zeros(3)                          # (origin: hand Euler 0, 0, 0, ENV:1040)
            else:
                eul = scenarios.hand_euler_for(o, self.np_random if with_noise == "zero-mean" else None)
            self.hand_euler[e] = eul
            hq[:, k] = euler_to_quat(eul)
            if noisy is not None:
                q[9:12, k] = noisy[row, :3] if row >= 0 else [0.0, 0.0, noisy[0, 2]]
                self.orientation_idx[e] = row
            elif start_pos is not None:
                sp = np.asarray(start_pos[k], dtype=np.float64)
                if len(sp) == 3:
                    q[9:12, k] = sp
                elif len(sp) == 2:                                                    # z = _get_obj_size()[-1] (the observation stores twice that)
                    q[9:12, k] = [sp[0], sp[1], self._obj_size_last(shape)]
                elif len(sp) == 9:
                    q[9:12, k] = sp[6:9]
                else:
                    raise ValueError("reset: a start_pos row has 3, 2 or 9 values")
                self.orientation_idx[e] = -1
            elif scenarios.has_start_table(shape, o, mode):
                tab = scenarios.start_coord_table(shape, o, mode)
                row = pick_row(tab)
                q[9:12, k] = tab[row] if row >= 0 else [0.0, 0.0, tab[0][2]]
                self.orientation_idx[e] = row
            else:
                # no coordinate file for this (shape, orientation) in the reference (Normal/BowlS ...): its empty-file rule (ENV:1243-1249, 821-849)
                q[9:12, k] = scenarios.fallback_start(shape, o, self.np_random)
                self.orientation_idx[e] = -1
            q[0:3, k] = scenarios.hand_slide_offsets(o, shape, self.hand_offsets)
            if start_pos is not None and len(start_pos[k]) == 9:
                sp = np.asarray(start_pos[k], dtype=np.float64)
                q[0:3, k] = sp[0:3]
                q[[3, 5, 7], k] = sp[3:6]
            self.obj_coords[e] = q[9:12, k]                                   # what the reference records: the COMMANDED point (ENV:1394)
            q[9:12, k] = scenarios.reset_body_position(shape, q[9:12, k])     # ... and where its 5 cm correction leaves the body (ENV:1379-1386)
            if qpos is not None:                                              # set_sim_state: the given joint vector, as it is
                q[:, k] = np.asarray(qpos[k], dtype=np.float64)
            self.hand_quat[:, e] = hq[:, k]
        if self.sim is None:                                                  # host_only: the draw itself
            return {"qpos": q, "hand_quat": hq}
        t_ids = None if env_ids is None else torch.as_tensor(ids, dtype=torch.int32)
        obs = self.sim.reset(torch.as_tensor(q), torch.as_tensor(hq), t_ids, object_id=self.shape_id[ids].copy() if multi else None,
                             mass_friction=obj_params if (obj_params is not None and not isinstance(obj_params, (list, tuple))) else None)
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
        if self.sim is not None:
            self.sim.close()
