"""ctypes binding of libkinova_sim.so (include/kinova_sim.h).  PyTorch is plumbing here: it owns the
device tensors and the stream; every compute call goes through the C ABI into the HIP kernels.
There is no CPU path: constructing a KinovaSim without the library or without a GPU raises."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import torch

from . import build as _build

NQ, NV, NACT, NOBS, NINFO, NCON_MAX, CONTACT_STRIDE = 16, 15, 4, 82, 3, 24, 20
NCON_MAX_MG = 40
# Newton iteration cap per substep (ks_config.solver_iterations).  MuJoCo's own cap is 100 with an early exit; the kernels exit
# as soon as the step is below 1e-5 of the solution or crossed no constraint row.  Over 13 200 grasp / lift substeps (3 hand
# poses, 14 shapes) no problem needs more than 10 iterations, while a cap of 6 - rounds 1 and 2 - truncated 2.3 % of the
# substeps with up to 20 % error in qacc (profiles/r03_solver_cap.txt).  A substep that still ends at the cap sets status bit 8.
SOLVER_ITERATIONS = 20
ASSETS = Path(__file__).resolve().parent / "assets"


class KsConfig(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("frame_skip", C.c_int32), ("horizon", C.c_int32), ("solver_iterations", C.c_int32),
                ("precision", C.c_int32), ("auto_reset", C.c_int32), ("obs_env_major", C.c_int32), ("envs_per_wave", C.c_int32),
                ("contact_tap", C.c_int32), ("pair_memory", C.c_int32), ("reserved", C.c_int32 * 2)]


class KrRing(C.Structure):
    """kr_ring (include/kinova_rollout.h): one device episode ring"""
    _fields_ = [("count", C.c_void_p), ("head", C.c_void_p), ("capacity", C.c_int32), ("ep_len", C.c_void_p), ("ep_state", C.c_void_p),
                ("ep_next", C.c_void_p), ("ep_action", C.c_void_p), ("ep_reward", C.c_void_p), ("ep_not_done", C.c_void_p)]


class KsRolloutArgs(C.Structure):
    """ks_rollout_args (include/kinova_sim.h)"""
    _fields_ = ([("actor_pub", C.c_void_p), ("actor_ver", C.c_void_p), ("actor_stride", C.c_int64)] +
                [(k, C.c_int64) for k in ("off_w1", "off_b1", "off_w2", "off_b2", "off_w3", "off_b3")] +
                [("h1", C.c_int32), ("h2", C.c_int32), ("sigma", C.c_float), ("max_action", C.c_float), ("skip_steps", C.c_int32),
                 ("with_replay", C.c_int32), ("seed", C.c_uint64)] +
                [(k, C.c_void_p) for k in ("obs", "prev_obs", "has_prev", "ready", "lifting", "t", "steps_total", "action", "action_t", "reward_out",
                                           "done_out", "sim_obs", "sim_reward", "sim_done", "sim_info", "sim_final_obs")] +
                [("horizon", C.c_int32), ("n_steps", C.c_int32)] +
                [(k, C.c_void_p) for k in ("cur_state", "cur_next", "cur_action", "cur_reward", "cur_not_done", "cur_len", "cur_sel", "pub_len", "counters")] +
                [("budget_ticks", C.c_int64)])


EXPORTS = ["ks_default_config", "ks_create", "ks_destroy", "ks_last_error", "ks_load_model", "ks_load_models", "ks_reset", "ks_reset_objects", "ks_step",
           "ks_get_state", "ks_set_state", "ks_set_env_params", "ks_substep", "ks_rollout", "ks_rollout_plan", "ks_obs_from_snapshot", "ks_kernel_time", "ks_version"]
# include/kinova_rollout.h
ROLLOUT_EXPORTS = ["kr_select_action", "kr_store_transition", "kr_rank_episodes", "kr_wait_min", "kr_wait_min_counted", "kr_commit_episodes", "kr_advance_ring",
                   "kr_sample_windows", "kr_sample_windows_draw", "kr_sample_windows_mixed", "kr_xchg_create", "kr_xchg_connect", "kr_xchg_allreduce_mean", "kr_xchg_status",
                   "kr_xchg_destroy", "kr_critic_grad", "kr_update_prologue", "kr_relu_backward", "kr_sigmoid_scale_backward", "kr_adam_step", "kr_soft_update",
                   "kr_mlp3_forward", "kr_mlp3_forward_shadow", "kr_mlp3_forward_split", "kr_mlp3_backward_shadow", "kr_mlp3_backward_split", "kr_weight_grad_shadow",
                   "kr_actor_select"]

_lib = None
_lib_mg = None


def load_library(path: Path | None = None, multi_geom: bool = False):
    """Load libkinova_sim.so - or, multi_geom=True, libkinova_sim_mg.so: the same simulator C ABI (include/kinova_sim.h) compiled with
    the capacities of the multi-geom objects (Bottle / TBottle / Bowl / RBowl: csrc/ks_model.h).  Must have been built
    (kinovagrasping_amd.build.build()).  Fails loudly."""
    global _lib, _lib_mg
    if multi_geom:
        if _lib_mg is None:
            import os
            mpath = Path(path) if path else Path(os.environ.get("KS_LIB_MG", _build.LIB_MG))
            if not mpath.exists():
                raise RuntimeError(f"{mpath} is missing: build it with kinovagrasping_amd.build.build() (hipcc, gfx950, -DKS_MULTI_GEOM). "
                                   "There is no fallback implementation.")
            _lib_mg = _bind(C.CDLL(str(mpath)))
        return _lib_mg
    if _lib is not None:
        return _lib
    import os
    path = Path(path) if path else Path(os.environ.get("KS_LIB", _build.LIB))
    if not path.exists():
        raise RuntimeError(f"{path} is missing: build it with kinovagrasping_amd.build.build() (hipcc, gfx950). "
                           "There is no fallback implementation.")
    _lib = _bind(C.CDLL(str(path)))
    return _lib


def blob_is_multi_geom(blob: bytes) -> bool:
    """does the model blob hold a multi-geom object (welded pieces beside `object`: more than the nine geoms of the fixed topology)?"""
    from .model_compiler import blob_record_shape
    return blob_record_shape(blob, "geom_body")[0] > 9


def blob_needs_mg_library(blob: bytes) -> bool:
    """a multi-geom object, or a single-geom one whose hull has more vertices than the standard library keeps in LDS (1024: the Lemon stand-in has 2434)"""
    from .model_compiler import blob_record_shape
    return blob_is_multi_geom(blob) or blob_record_shape(blob, "mesh3_vert")[0] > 1024


def _bind(L):
    vp, i32p = C.c_void_p, C.c_void_p
    L.ks_default_config.argtypes = [C.POINTER(KsConfig)]
    L.ks_create.argtypes = [C.POINTER(KsConfig), C.c_int, C.POINTER(vp)]
    L.ks_destroy.argtypes = [vp]
    L.ks_last_error.argtypes = [vp]
    L.ks_last_error.restype = C.c_char_p
    L.ks_load_model.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.ks_load_models.argtypes = [vp, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]
    L.ks_reset.argtypes = [vp, i32p, C.c_int32, vp, vp, vp, vp]
    L.ks_reset_objects.argtypes = [vp, i32p, C.c_int32, vp, vp, vp, vp, vp, vp]
    L.ks_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.ks_get_state.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.ks_set_state.argtypes = [vp, vp, vp, vp, vp]
    L.ks_set_env_params.argtypes = [vp, vp, vp, vp]
    L.ks_substep.argtypes = [vp, vp, vp]
    L.ks_rollout.argtypes = [vp, C.c_int32, C.POINTER(KsRolloutArgs), vp]
    L.ks_obs_from_snapshot.argtypes = [vp] * 8
    L.ks_kernel_time.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.ks_rollout_plan.argtypes = [vp, i32p, i32p, i32p]
    i32, f32 = C.c_int32, C.c_float
    L.kr_select_action.argtypes = [i32] + [vp] * 7 + [f32, f32, i32] + [vp] * 4
    L.kr_store_transition.argtypes = [i32] * 5 + [vp] * 21
    L.kr_rank_episodes.argtypes = [i32, vp, vp, vp, vp]
    L.kr_wait_min.argtypes = [vp, i32, C.c_int64, C.c_double, vp]
    L.kr_wait_min_counted.argtypes = [vp, i32, C.c_int64, C.c_double, vp, vp]
    L.kr_commit_episodes.argtypes = [i32, i32, i32] + [vp] * 16
    L.kr_advance_ring.argtypes = [i32, i32] + [vp] * 6
    L.kr_sample_windows.argtypes = [i32, i32, i32, vp, vp, i32] + [vp] * 15
    L.kr_sample_windows_draw.argtypes = [i32, i32, i32, vp, vp, i32, vp, C.c_uint64, vp] + [vp] * 13
    L.kr_sample_windows_mixed.argtypes = [i32, i32, i32, i32, C.POINTER(KrRing), C.POINTER(KrRing), vp, vp, C.c_uint64, vp] + [vp] * 7 + [vp]
    L.kr_xchg_create.argtypes = [C.POINTER(vp), i32, i32, C.c_int64, vp]
    L.kr_xchg_connect.argtypes = [vp, C.c_char_p]
    L.kr_xchg_allreduce_mean.argtypes = [vp, vp, C.c_int64, vp]
    L.kr_xchg_status.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.kr_xchg_destroy.argtypes = [vp]
    L.kr_xchg_destroy.restype = None
    i64 = C.c_int64
    L.kr_critic_grad.argtypes = [i32, i32] + [vp] * 6 + [f32, vp, vp, vp]
    L.kr_update_prologue.argtypes = [i32, i32, vp, vp, vp, vp, vp, i32, vp]
    L.kr_relu_backward.argtypes = [i64, vp, vp, vp]
    L.kr_sigmoid_scale_backward.argtypes = [i64, vp, f32, vp, vp]
    L.kr_adam_step.argtypes = [i64, vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, vp]
    L.kr_soft_update.argtypes = [i64, vp, vp, f32, vp, i32, vp]
    L.kr_mlp3_forward.argtypes = [i32] * 6 + [vp, i32, vp, i32] + [vp] * 6 + [i32, f32, vp, vp, vp, vp]
    L.kr_mlp3_forward_shadow.argtypes = L.kr_mlp3_forward.argtypes
    L.kr_mlp3_forward_split.argtypes = [i32] * 6 + [vp, i32, vp, i32] + [vp] * 6 + [i32, f32, vp, vp, vp, vp, C.c_int64, i32, vp]
    L.kr_mlp3_backward_shadow.argtypes = [i32] * 5 + [vp] * 8 + [i32, i32, vp, f32, vp, vp]
    L.kr_mlp3_backward_split.argtypes = [i32] * 5 + [vp] * 8 + [i32, i32, vp, f32, vp, vp, C.c_int64, i32, vp]
    L.kr_weight_grad_shadow.argtypes = [i32] * 4 + [vp, vp, i32, vp, i32, i32, vp, vp, vp, vp]
    L.kr_actor_select.argtypes = [i32] * 3 + [vp] * 12 + [C.c_uint64, vp, f32, f32, i32] + [vp] * 5
    return L


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class KinovaSim:
    """N batched envs on one GPU.  All tensors are torch CUDA tensors, field-major [K, N] unless noted."""

    def __init__(self, n_envs: int, model: str | bytes = "CubeS", device: int | torch.device = 0, precision: int = 32,
                 frame_skip: int = 15, horizon: int = 30, solver_iterations: int = SOLVER_ITERATIONS, auto_reset: bool = False,
                 obs_env_major: bool = True, envs_per_wave: int = 0, contact_tap: bool = False, pair_memory: bool = True):
        if not torch.cuda.is_available():
            raise RuntimeError("KinovaSim needs a HIP GPU (torch.cuda.is_available() is False); there is no CPU path")
        from .model_compiler import load_model_blob
        as_blob = lambda m: bytes(m) if isinstance(m, (bytes, bytearray)) else bytes(load_model_blob(m, ASSETS))
        self.models = list(model) if isinstance(model, (list, tuple)) else [model]
        blobs = [as_blob(m) for m in self.models]
        # a context with a multi-geom object (welded pieces: Bottle / TBottle / Bowl / RBowl) runs on the library built with those
        # capacities; it holds single-geom objects as well (slower than the standard library: hull tables in L2, not LDS)
        self.multi_geom = any(blob_needs_mg_library(b) for b in blobs)
        self.lib = load_library(multi_geom=self.multi_geom)
        self.ncon_max = NCON_MAX_MG if self.multi_geom else NCON_MAX           # contact records of the parity tap (include/kinova_sim.h)
        self.device = torch.device("cuda", device if isinstance(device, int) else (device.index or 0))
        self.n_envs = int(n_envs)
        self.dtype = torch.float32 if precision == 32 else torch.float64
        self.obs_env_major = bool(obs_env_major)
        cfg = KsConfig()
        self.lib.ks_default_config(C.byref(cfg))
        cfg.n_envs, cfg.frame_skip, cfg.horizon, cfg.solver_iterations = self.n_envs, frame_skip, horizon, solver_iterations
        cfg.precision, cfg.auto_reset, cfg.obs_env_major = precision, int(auto_reset), int(obs_env_major)
        cfg.envs_per_wave, cfg.contact_tap, cfg.pair_memory = int(envs_per_wave), int(contact_tap), int(pair_memory)
        self.cfg = cfg
        self.ctx = C.c_void_p()
        rc = self.lib.ks_create(C.byref(cfg), self.device.index, C.byref(self.ctx))
        if rc != 0:
            raise RuntimeError(f"ks_create failed ({rc}): {self.lib.ks_last_error(None).decode()}")
        if isinstance(model, (list, tuple)):
            # mixed-object context (BASELINE config 5): object k of reset(object_id=...) is model[k]
            arr = (C.c_char_p * len(blobs))(*blobs)
            sizes = (C.c_size_t * len(blobs))(*[len(b) for b in blobs])
            self._check(self.lib.ks_load_models(self.ctx, len(blobs), arr, sizes))
        else:
            self._check(self.lib.ks_load_model(self.ctx, blobs[0], len(blobs[0])))
        N, dt, dev = self.n_envs, self.dtype, self.device
        self.obs = torch.zeros((N, NOBS) if obs_env_major else (NOBS, N), dtype=dt, device=dev)
        self.final_obs = torch.zeros_like(self.obs)
        self.reward = torch.zeros(N, dtype=dt, device=dev)
        self.done = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.info = torch.zeros((NINFO, N), dtype=dt, device=dev)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"libkinova_sim error {rc}: {self.lib.ks_last_error(self.ctx).decode()}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.ks_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, qpos0: torch.Tensor, hand_quat: torch.Tensor, env_ids: torch.Tensor | None = None, object_id=None, mass_friction=None):
        """qpos0 [16, n], hand_quat [4, n]; env_ids int32 [n] or None (all envs); object_id int32 [n] (index into the
        context's model list) and mass_friction [2, n] (object mass, object-hand friction) optional (ks_reset_objects).
        Returns the obs buffer."""
        qpos0 = qpos0.to(self.device, self.dtype).contiguous()
        hand_quat = hand_quat.to(self.device, self.dtype).contiguous()
        n = qpos0.shape[1]
        ids = None if env_ids is None else env_ids.to(self.device, torch.int32).contiguous()
        if object_id is None and mass_friction is None:
            self._check(self.lib.ks_reset(self.ctx, _ptr(ids), n, _ptr(qpos0), _ptr(hand_quat), _ptr(self.obs), self._stream()))
            self._keep = (qpos0, hand_quat, ids)
            return self.obs
        oid = None if object_id is None else torch.as_tensor(object_id).to(self.device, torch.int32).contiguous()
        mf = None if mass_friction is None else torch.as_tensor(mass_friction).to(self.device, self.dtype).contiguous()
        if oid is not None and oid.numel() != n or mf is not None and tuple(mf.shape) != (2, n):
            raise ValueError("reset: object_id [n], mass_friction [2, n]")
        self._check(self.lib.ks_reset_objects(self.ctx, _ptr(ids), n, _ptr(qpos0), _ptr(hand_quat), _ptr(oid), _ptr(mf), _ptr(self.obs), self._stream()))
        self._keep = (qpos0, hand_quat, ids, oid, mf)
        return self.obs

    def step(self, action: torch.Tensor):
        """action [4, N].  Returns (obs, reward, done, info) views of the context's output buffers."""
        action = action.to(self.device, self.dtype).contiguous()
        self._check(self.lib.ks_step(self.ctx, _ptr(action), _ptr(self.obs), _ptr(self.reward), _ptr(self.done), _ptr(self.info),
                                     _ptr(self.final_obs), self._stream()))
        self._keep_a = action
        return self.obs, self.reward, self.done, self.info

    def rollout(self, n_iter: int, args: "KsRolloutArgs"):
        """n_iter free-running env-steps of every env in one launch (ks_rollout: in-kernel actor + step + replay write)"""
        self._check(self.lib.ks_rollout(self.ctx, int(n_iter), C.byref(args), self._stream()))

    def substep(self, ctrl: torch.Tensor):
        ctrl = ctrl.to(self.device, self.dtype).contiguous()
        self._check(self.lib.ks_substep(self.ctx, _ptr(ctrl), self._stream()))
        self._keep_c = ctrl

    def obs_from_snapshot(self, snap: torch.Tensor, rays: torch.Tensor):
        """snap [105, N] (poses of bodies 2..9, then the 9 jointpos sensors), rays [17, N] -> (obs, reward, done, info) of
        that engine state (ks_obs_from_snapshot: _get_obs + _get_reward without stepping)."""
        snap = snap.to(self.device, self.dtype).contiguous()
        rays = rays.to(self.device, self.dtype).contiguous()
        if tuple(snap.shape) != (105, self.n_envs) or tuple(rays.shape) != (17, self.n_envs):
            raise ValueError("obs_from_snapshot: snap [105, N], rays [17, N]")
        self._check(self.lib.ks_obs_from_snapshot(self.ctx, _ptr(snap), _ptr(rays), _ptr(self.obs), _ptr(self.reward), _ptr(self.done),
                                                  _ptr(self.info), self._stream()))
        self._keep_s = (snap, rays)
        return self.obs, self.reward, self.done, self.info

    def get_state(self, contacts: bool = False):
        N, dt, dev = self.n_envs, self.dtype, self.device
        out = dict(qpos=torch.empty((NQ, N), dtype=dt, device=dev), qvel=torch.empty((NV, N), dtype=dt, device=dev),
                   qacc_warmstart=torch.empty((NV, N), dtype=dt, device=dev),
                   ncon=torch.empty(N, dtype=torch.int32, device=dev), status=torch.empty(N, dtype=torch.int32, device=dev))
        con = torch.empty((self.ncon_max * CONTACT_STRIDE, N), dtype=dt, device=dev) if contacts else None
        self._check(self.lib.ks_get_state(self.ctx, _ptr(out["qpos"]), _ptr(out["qvel"]), _ptr(out["qacc_warmstart"]), _ptr(con),
                                          _ptr(out["ncon"]), _ptr(out["status"]), self._stream()))
        if contacts:
            out["contact"] = con.view(self.ncon_max, CONTACT_STRIDE, N)
        return out

    def set_state(self, qpos=None, qvel=None, qacc_warmstart=None):
        ts = [None if t is None else t.to(self.device, self.dtype).contiguous() for t in (qpos, qvel, qacc_warmstart)]
        self._check(self.lib.ks_set_state(self.ctx, _ptr(ts[0]), _ptr(ts[1]), _ptr(ts[2]), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()

    def set_env_params(self, obj_mass=None, obj_mu=None):
        """Per-env object mass [N] (kg) and object-hand friction [N] (BASELINE config 5); None leaves a parameter as is."""
        ts = [None if t is None else torch.as_tensor(t).to(self.device, self.dtype).contiguous() for t in (obj_mass, obj_mu)]
        for t in ts:
            if t is not None and t.numel() != self.n_envs:
                raise ValueError("set_env_params: one value per env")
        self._check(self.lib.ks_set_env_params(self.ctx, _ptr(ts[0]), _ptr(ts[1]), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()

    ROLLOUT_PLANS = ("waves", "workgroups", "queue", "runs", "round-robin")

    def rollout_plan(self):
        """(plan, groups, workgroups): how ks_rollout schedules this context's 16-env groups (include/kinova_sim.h: ks_rollout_plan) - "waves" / "workgroups":
        one group per persistent workgroup (waves free / joined by barriers); "queue": more groups than workgroups, from a ready queue; "runs" /
        "round-robin": a fixed deal - the launch runs at the pace of the workgroup with the most groups."""
        m, g, w = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._check(self.lib.ks_rollout_plan(self.ctx, C.byref(m), C.byref(g), C.byref(w)))
        return self.ROLLOUT_PLANS[m.value], g.value, w.value

    def kernel_time(self, reset: bool = False):
        """(average ms of the env-step kernel measured with HIP events on the launch stream, launches)."""
        ms, n = C.c_double(0), C.c_int64(0)
        self._check(self.lib.ks_kernel_time(self.ctx, int(reset), C.byref(ms), C.byref(n)))
        return ms.value, n.value
