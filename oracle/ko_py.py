"""ctypes binding of the CPU oracle (oracle/libko_oracle.so).  TEST INFRASTRUCTURE ONLY:
import this from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, nowhere else."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
NQ, NV, NU, NBODY, NGEOM, NSITE, NSENSOR = 16, 15, 9, 10, 17, 17, 26      # NGEOM: capacity (ko.h KO_NGEOM); single-geom objects use the first 9
NCON_MAX = 40       # capacity (ko.h KO_NCON_MAX); Sim.ncon_max is the number kept: 24, multi-geom objects 40
NEFC_MAX = 3 + 9 + 4 * NCON_MAX
NOBS, NOBS_GLOBAL = 82, 74
d = C.c_double


class Contact(C.Structure):
    _fields_ = [("dist", d), ("pos", d * 3), ("frame", d * 9), ("mu", d * 2), ("margin", d),
                ("geom1", C.c_int), ("geom2", C.c_int)]


class Sim(C.Structure):
    _fields_ = [
        ("m", C.c_void_p), ("hand_quat", d * 4), ("solver", C.c_int), ("solver_iterations", C.c_int), ("ncon_max", C.c_int),
        ("qpos", d * NQ), ("qvel", d * NV), ("qacc_warmstart", d * NV), ("ctrl", d * NU),
        ("xpos", d * 3 * NBODY), ("xmat", d * 9 * NBODY), ("xipos", d * 3 * NBODY), ("ximat", d * 9 * NBODY),
        ("geom_xpos", d * 3 * NGEOM), ("geom_xmat", d * 9 * NGEOM),
        ("site_xpos", d * 3 * NSITE), ("site_xmat", d * 9 * NSITE),
        ("M", d * NV * NV), ("L", d * NV * NV),
        ("ncon", C.c_int), ("ncon_dropped", C.c_int), ("contact", Contact * NCON_MAX),
        ("nefc", C.c_int), ("efc_type", C.c_int * NEFC_MAX),
        ("efc_J", d * NV * NEFC_MAX), ("efc_pos", d * NEFC_MAX), ("efc_margin", d * NEFC_MAX),
        ("efc_R", d * NEFC_MAX), ("efc_aref", d * NEFC_MAX), ("efc_b", d * NEFC_MAX), ("efc_force", d * NEFC_MAX),
        ("sensordata", d * NSENSOR),
        ("qfrc_bias", d * NV), ("qfrc_passive", d * NV), ("qfrc_actuator", d * NV), ("qfrc_constraint", d * NV),
        ("qacc_smooth", d * NV), ("qacc", d * NV),
        ("mpr_calls", C.c_int), ("mpr_support_calls", C.c_int), ("newton_last_grad", d), ("newton_iters_used", C.c_int), ("rays_enabled", C.c_int),
        ("obj_mass", d), ("obj_mu", d),
        ("solver_tolerance", d), ("newton_last_step", d), ("newton_converged", C.c_int), ("narrow_phase", C.c_int),
    ]


class EnvInputs(C.Structure):
    _fields_ = [("palm_xpos", d * 3), ("palm_xmat", d * 9), ("finger_xpos", d * 3 * 6), ("obj_xpos", d * 3),
                ("link7_xpos", d * 3), ("site_xpos", d * 3 * NSITE), ("sensordata", d * NSENSOR), ("obj_size", d * 3)]


_lib = None


def build(force: bool = False) -> Path:
    so = HERE / "libko_oracle.so"
    srcs = [HERE / n for n in ("ko_model.c", "ko_physics.c", "ko_env.c", "ko.h")]
    if force or not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        subprocess.check_call(["make", "-C", str(HERE), "-s"])
    return so


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(build()))
        L = _lib
        L.ko_model_load.restype = C.c_void_p
        L.ko_model_load.argtypes = [C.c_char_p, C.c_size_t]
        L.ko_model_free.argtypes = [C.c_void_p]
        L.ko_sim_new.restype = C.POINTER(Sim)
        L.ko_sim_new.argtypes = [C.c_void_p, C.POINTER(d)]
        L.ko_sim_free.argtypes = [C.POINTER(Sim)]
        L.ko_set_state.argtypes = [C.POINTER(Sim), C.POINTER(d), C.POINTER(d), C.POINTER(d)]
        for f in (L.ko_forward, L.ko_step, L.ko_kinematics):
            f.argtypes = [C.POINTER(Sim)]
        L.ko_env_palm_transform.argtypes = [C.POINTER(d)] * 4
        L.ko_env_ctrl.argtypes = [C.POINTER(d), C.POINTER(d), C.c_int, C.POINTER(d)]
        L.ko_env_obs_local.argtypes = [C.POINTER(EnvInputs), C.POINTER(d)]
        L.ko_env_obs_global.argtypes = [C.POINTER(EnvInputs), C.POINTER(d)]
        L.ko_env_reward.argtypes = [d, C.POINTER(d), C.POINTER(C.c_int), C.POINTER(d)]
        L.ko_check_grasp.argtypes = [C.POINTER(d), C.POINTER(d)]
        L.ko_check_grasp.restype = C.c_int
        L.ko_env_inputs_from_sim.argtypes = [C.POINTER(Sim), C.POINTER(EnvInputs)]
        L.ko_env_step.argtypes = [C.POINTER(Sim), C.POINTER(d), C.c_int, C.c_int, C.POINTER(d), C.POINTER(d),
                                  C.POINTER(C.c_int), C.POINTER(d)]
        L.ko_env_reset.argtypes = [C.POINTER(Sim), C.POINTER(d), C.POINTER(d)]
        L.ko_sizeof_sim.restype = C.c_size_t
        assert L.ko_sizeof_sim() == C.sizeof(Sim), (L.ko_sizeof_sim(), C.sizeof(Sim))
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(d))


def _np(carr):
    return np.ctypeslib.as_array(carr)


class OracleModel:
    def __init__(self, blob: bytes):
        self._blob = bytes(blob)
        self.ptr = lib().ko_model_load(self._blob, len(self._blob))
        if not self.ptr:
            raise ValueError("ko_model_load failed")

    def __del__(self):
        try:
            if getattr(self, "ptr", None) and _lib is not None:
                _lib.ko_model_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class OracleSim:
    """One fp64 env.  Arrays returned by properties are live views into the C struct."""

    def __init__(self, model: OracleModel, hand_quat, solver_iterations: int = 8, solver: int = 0, ncon_max: int | None = None):
        """ncon_max: contacts kept per substep (ko_sim_new: 24, 40 for multi-geom objects = the two product libraries' NCON_MAX).  A test that
        runs a SINGLE-geom object in a context of libkinova_sim_mg.so passes 40: that library keeps 40 for every model it loads."""
        self.model = model
        hq = np.ascontiguousarray(hand_quat, dtype=np.float64)
        self.p = lib().ko_sim_new(model.ptr, _p(hq))
        self.s = self.p.contents
        self.s.solver_iterations = solver_iterations
        self.s.solver = solver
        if ncon_max is not None:
            assert 0 < ncon_max <= NCON_MAX
            self.s.ncon_max = ncon_max

    def __del__(self):
        try:
            if getattr(self, "p", None) and _lib is not None:
                _lib.ko_sim_free(self.p)
                self.p = None
        except Exception:
            pass

    def view(self, name):
        return _np(getattr(self.s, name))

    def set_state(self, qpos, qvel=None, warm=None):
        qpos = np.ascontiguousarray(qpos, dtype=np.float64)
        qv = None if qvel is None else np.ascontiguousarray(qvel, dtype=np.float64)
        wa = None if warm is None else np.ascontiguousarray(warm, dtype=np.float64)
        lib().ko_set_state(self.p, _p(qpos), None if qv is None else _p(qv), None if wa is None else _p(wa))

    def forward(self):
        lib().ko_forward(self.p)

    def step(self, ctrl=None):
        if ctrl is not None:
            self.view("ctrl")[:] = ctrl
        lib().ko_step(self.p)

    def env_reset(self, qpos0):
        q = np.ascontiguousarray(qpos0, dtype=np.float64)
        obs = np.zeros(NOBS)
        lib().ko_env_reset(self.p, _p(q), _p(obs))
        return obs

    def env_step(self, action, frame_skip=15):
        a = np.ascontiguousarray(action, dtype=np.float64)
        obs = np.zeros(NOBS)
        rew = d(0)
        done = C.c_int(0)
        info = np.zeros(3)
        lib().ko_env_step(self.p, _p(a), len(a), frame_skip, _p(obs), C.byref(rew), C.byref(done), _p(info))
        return obs, rew.value, bool(done.value), info

    def contacts(self):
        out = []
        for i in range(self.s.ncon):
            c = self.s.contact[i]
            out.append(dict(dist=c.dist, pos=np.array(c.pos[:]), frame=np.array(c.frame[:]), geom1=c.geom1, geom2=c.geom2,
                            mu=np.array(c.mu[:])))
        return out


def _contact_forces(self):
    """[ncon, 3] constraint force of every contact of the last forward pass in its own frame (normal, tangent 1,
    tangent 2), recovered from the pyramid rows of efc_force as MuJoCo's mj_contactForce does: normal = sum of the four
    edge forces, tangent k = mu_k * (f+ - f-).  Contacts outside their margin have no rows (zero force)."""
    s = self.s
    ty = _np(s.efc_type)[:s.nefc]
    f = _np(s.efc_force)[:s.nefc]
    row = int((ty != 2).sum())
    out = np.zeros((s.ncon, 3))
    for i in range(s.ncon):
        c = s.contact[i]
        if c.dist >= c.margin:
            continue
        e = f[row:row + 4]
        out[i] = [e.sum(), c.mu[0] * (e[0] - e[1]), c.mu[1] * (e[2] - e[3])]
        row += 4
    assert row == s.nefc
    return out


OracleSim.contact_forces = _contact_forces
GEOM_BODY = [0, 2, 3, 4, 5, 6, 7, 8, 9] + [9] * 8        # body of geom g (ground, palm, f1_prox, f1_dist, ..., object, welded object pieces)


def env_obs_from_inputs(inputs: dict):
    """Pure env-layer functions on fake-sim inputs (golden-vector pinning)."""
    ei = EnvInputs()
    for k in ("palm_xpos", "palm_xmat", "obj_xpos", "link7_xpos", "sensordata", "obj_size"):
        v = np.asarray(inputs[k], dtype=np.float64).ravel()
        getattr(ei, k)[:] = v.tolist()
    fx = np.asarray(inputs["finger_xpos"], dtype=np.float64).reshape(6, 3)
    sx = np.asarray(inputs["site_xpos"], dtype=np.float64).reshape(NSITE, 3)
    for i in range(6):
        ei.finger_xpos[i][:] = fx[i].tolist()
    for i in range(NSITE):
        ei.site_xpos[i][:] = sx[i].tolist()
    ol = np.zeros(NOBS)
    og = np.zeros(NOBS_GLOBAL)
    lib().ko_env_obs_local(C.byref(ei), _p(ol))
    lib().ko_env_obs_global(C.byref(ei), _p(og))
    return ol, og


def env_ctrl(palm_xpos, palm_xmat, action):
    Tfw = np.zeros(16)
    wrist = np.zeros(3)
    px = np.ascontiguousarray(palm_xpos, dtype=np.float64)
    pm = np.ascontiguousarray(palm_xmat, dtype=np.float64).ravel()
    lib().ko_env_palm_transform(_p(px), _p(pm), _p(Tfw), _p(wrist))
    a = np.ascontiguousarray(action, dtype=np.float64)
    ctrl = np.zeros(9)
    lib().ko_env_ctrl(_p(Tfw), _p(a), len(a), _p(ctrl))
    return Tfw.reshape(4, 4), wrist, ctrl


def env_reward(z):
    rew = d(0)
    done = C.c_int(0)
    info = np.zeros(3)
    lib().ko_env_reward(float(z), C.byref(rew), C.byref(done), _p(info))
    return rew.value, bool(done.value), info


def check_grasp(old, new):
    o = np.ascontiguousarray(old, dtype=np.float64)
    n = np.ascontiguousarray(new, dtype=np.float64)
    return bool(lib().ko_check_grasp(_p(o), _p(n)))
