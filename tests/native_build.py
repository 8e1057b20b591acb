"""Builds tests/native/libks_lanecheck.so (host build of the kernel source, TEST ONLY)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from kinovagrasping_amd.sim import SOLVER_ITERATIONS

HERE = Path(__file__).resolve().parent / "native"
CSRC = Path(__file__).resolve().parents[1] / "kinovagrasping_amd" / "csrc"
dp = C.POINTER(C.c_double)


def lanecheck_lib(multi_geom: bool = False):
    """multi_geom: the kernel source compiled with -DKS_MULTI_GEOM (the topology capacities of libkinova_sim_mg.so)"""
    so, src = HERE / ("libks_lanecheck_mg.so" if multi_geom else "libks_lanecheck.so"), HERE / "ks_lanecheck.cpp"
    deps = [src] + sorted(CSRC.glob("*.h"))
    if not so.exists() or any(d.stat().st_mtime > so.stat().st_mtime for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared"] + (["-DKS_MULTI_GEOM"] if multi_geom else []) + ["-o", str(so), str(src)])
    L = C.CDLL(str(so))
    L.lc_create.restype = C.c_void_p
    L.lc_create.argtypes = [C.c_char_p, C.c_size_t]
    L.lc_destroy.argtypes = [C.c_void_p]
    L.lc_substep.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, C.c_int, C.POINTER(C.c_int), dp]
    L.lc_env_step.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, C.c_int, C.c_int, dp, dp, C.POINTER(C.c_int), dp]
    L.lc_reset_obs.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, dp, C.POINTER(C.c_int), dp]
    return L


def P(a):
    return a.ctypes.data_as(dp)


class Lane:
    """one lane of the kernel source on the CPU, fp32 or fp64"""

    def __init__(self, blob: bytes, prec: int, iters: int = SOLVER_ITERATIONS, multi_geom: bool = False):
        self.L = lanecheck_lib(multi_geom)
        self.h = self.L.lc_create(blob, len(blob))
        assert self.h, "lc_create failed (see stderr)"
        self.prec, self.iters = prec, iters

    def substep(self, qpos, qvel, warm, ctrl, hq):
        a, b, c = (np.array(x, dtype=np.float64) for x in (qpos, qvel, warm))
        nc = C.c_int(0)
        ncm = self.L.lc_ncon_max()
        con = np.zeros(ncm * 20)
        st = self.L.lc_substep(self.h, self.prec, P(a), P(b), P(c), P(np.array(ctrl, dtype=np.float64)), P(np.array(hq, dtype=np.float64)),
                               self.iters, C.byref(nc), P(con))
        return a, b, c, nc.value, con.reshape(ncm, 20), st

    def env_step(self, qpos, qvel, warm, hq, act, frame_skip=15):
        a, b, c = (np.array(x, dtype=np.float64) for x in (qpos, qvel, warm))
        obs, rays, rew, done = np.zeros(82), np.zeros(17), C.c_double(0), C.c_int(0)
        st = self.L.lc_env_step(self.h, self.prec, P(a), P(b), P(c), P(np.array(hq, dtype=np.float64)), P(np.array(act, dtype=np.float64)),
                                frame_skip, self.iters, P(obs), C.byref(rew), C.byref(done), P(rays))
        return a, b, c, obs, rew.value, bool(done.value), rays, st

    def reset_obs(self, qpos0, hq):
        a, b, c = np.array(qpos0, dtype=np.float64), np.zeros(15), np.zeros(15)
        obs, rays, rew, done = np.zeros(82), np.zeros(17), C.c_double(0), C.c_int(0)
        self.L.lc_reset_obs(self.h, self.prec, P(a), P(b), P(c), P(np.array(hq, dtype=np.float64)), P(obs), C.byref(rew), C.byref(done), P(rays))
        return obs, rays
