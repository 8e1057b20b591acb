#!/usr/bin/env python3
"""CPU study (test infrastructure: the oracle is the checker): the multi-geom objects' long-horizon parity on the HOST lane of the kernel source
(-DKS_MULTI_GEOM), from the start states of tools/debug/mg_long_horizon.py: queries cold / with the lanes' pair memory (the GPU's ks_step), and
build variants.  usage: python tools/r06/mg_host_study.py [flags ...]"""
import ctypes as C, subprocess, sys
from concurrent.futures import ProcessPoolExecutor
import numpy as np
sys.path.insert(0, '.')
from tests import native_build
from kinovagrasping_amd import scenarios
from kinovagrasping_amd.vec_env import KinovaGripperVecEnv
from kinovagrasping_amd.sim import SOLVER_ITERATIONS

N_SUB, per = 200, 6
script = np.array([[0.0, 0.6, 0.5, 0.7]] * 9 + [[0.6, 0.5, 0.5, 0.5]] * 5)
shapes = ("BottleS", "BottleB", "TBottleS", "TBottleM", "BowlS", "BowlB", "RBowlS", "RBowlM")


def lib(name):
    def load(multi_geom=False):
        L = C.CDLL(f"/tmp/libks_lc_mg_{name}.so")
        L.lc_create.restype = C.c_void_p; L.lc_create.argtypes = [C.c_char_p, C.c_size_t]
        L.lc_substep.argtypes = [C.c_void_p, C.c_int, native_build.dp, native_build.dp, native_build.dp, native_build.dp, native_build.dp, C.c_int, C.POINTER(C.c_int), native_build.dp]
        L.lc_set_warm.argtypes = [C.c_void_p, C.c_int]
        return L
    return load


def run(args):
    sh, q0, hq, name, warm = args
    from oracle import ko_py as ko
    native_build.lanecheck_lib = lib(name)
    blob = scenarios.model_blob(sh)
    m = ko.OracleModel(blob)
    ref = ko.OracleSim(m, hq, solver_iterations=SOLVER_ITERATIONS); ref.s.rays_enabled = 0
    ref.env_reset(q0.copy())
    lane = native_build.Lane(blob, 6432 if warm >= 10 else 32, multi_geom=True)      # warm >= 10: ks_lanecheck.cpp substep_mixed, variant warm - 10
    if warm >= 10:
        lane.L.lc_set_mixed_variant(warm - 10)
    else:
        lane.L.lc_set_warm(lane.h, warm)
    st = (ref.view("qpos").copy(), ref.view("qvel").copy(), ref.view("qacc_warmstart").copy())
    rel = 0.0
    for k in range(N_SUB):
        if k % 15 == 0:
            ctrl = ko.env_ctrl(ref.view("geom_xpos").reshape(-1, 3)[1], ref.view("geom_xmat").reshape(-1, 9)[1], script[k // 15])[2]
        ref.step(ctrl)
        qp, qv, qw, nc, con, status = lane.substep(*st, ctrl, hq)
        st = (qp, qv, qw)
        qo = ref.view("qpos")
        rel = np.abs(qp - qo).max() / max(1e-3, np.abs(qo).max())
    return sh, rel


if __name__ == "__main__":
    variants = {"default": []}
    for a in sys.argv[1:]:
        variants[a] = a.split(",")
    for name, flags in variants.items():
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DKS_MULTI_GEOM"] + flags + ["-o", f"/tmp/libks_lc_mg_{name}.so", str(native_build.HERE / "ks_lanecheck.cpp")])
    starts = {}
    for sh in shapes:
        env = KinovaGripperVecEnv(per, sh, seed=11, host_only=True)
        st = env.reset([sh], "normal", with_noise=False)
        starts[sh] = (st["qpos"], st["hand_quat"])
    for name in variants:
        for warm in (0, 1, 13, 12):
            jobs = [(sh, starts[sh][0][:, i].copy(), starts[sh][1][:, i].copy(), name, warm) for sh in shapes for i in range(per)]
            with ProcessPoolExecutor(8) as ex:
                res = list(ex.map(run, jobs, chunksize=2))
            per_sh = {sh: sum(1 for s, r in res if s == sh and r <= 1e-4) for sh in shapes}
            print(f"{name:28s} { {0: 'cold', 1: 'pair memory', 13: 'hull pairs from an fp64 collision stage', 12: 'whole collision stage fp64'}[warm]:42s}: {sum(per_sh.values())} of {len(res)} within 1e-4 at substep {N_SUB}   {per_sh}", flush=True)
