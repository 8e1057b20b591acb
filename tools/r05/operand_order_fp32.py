#!/usr/bin/env python3
"""Round-5 study (CPU): does the operand order of the convex queries change how well the fp32 kernel lane follows the fp64 oracle?
BASELINE config-2 envs (CubeS, random +-0.8 actions) run on the oracle; at every substep the fp32 lane (kernel source on the host) takes ONE
substep from the oracle's state.  Both sides built with the same order: object first (MuJoCo's, round 5) / hand geom first (rounds 1-4).
usage: python tools/r05/operand_order_fp32.py [n_envs] [env_steps]"""
import sys, subprocess, ctypes
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from kinovagrasping_amd import scenarios
from kinovagrasping_amd.sim import SOLVER_ITERATIONS
from tests import native_build
n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
blob = scenarios.model_blob(sys.argv[3] if len(sys.argv) > 3 else "CubeS")
q0, hq = scenarios.config2_states(n_envs)
acts = scenarios.config_actions(n_envs, n_steps)

def build_lane(obj_first):
    so = Path(f"/tmp/libks_lanecheck_of{obj_first}.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", f"-DKS_OBJ_FIRST={obj_first}", f"-I{native_build.CSRC}", "-o", str(so), str(native_build.HERE / "ks_lanecheck.cpp")])
    return so

for obj_first in (1, 0):
    so = build_lane(obj_first)
    native_build.lanecheck_lib = lambda multi_geom=False, so=so: _load(so)
    def _load(so):
        L = ctypes.CDLL(str(so)); dp = native_build.dp; C = ctypes
        L.lc_create.restype = C.c_void_p; L.lc_create.argtypes = [C.c_char_p, C.c_size_t]
        L.lc_substep.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, C.c_int, C.POINTER(C.c_int), dp]
        return L
    lane = native_build.Lane(blob, 32)
    if obj_first == 1:
        from oracle import ko_py as ko
    else:
        sys.path.insert(0, "/tmp/dbg")
        for k in [k for k in sys.modules if k.startswith("oracle")]: del sys.modules[k]
        from oracle import ko_py as ko
        ctypes.c_int.in_dll(ko.lib(), "ko_dbg_obj_first").value = 0
    m = ko.OracleModel(blob)
    errs, ncm, touching = [], 0, 0
    for e in range(n_envs):
        s = ko.OracleSim(m, hq[:, e], solver_iterations=SOLVER_ITERATIONS); s.s.rays_enabled = 0
        s.env_reset(q0[:, e])
        for t in range(n_steps):
            a = acts[t][:, e]
            ctrl = np.zeros(9)
            import ctypes as C
            ko.lib().ko_env_ctrl(s.view("geom_xmat")[9:18].ctypes.data_as(C.POINTER(C.c_double)), np.ascontiguousarray(a).ctypes.data_as(C.POINTER(C.c_double)), 4, ctrl.ctypes.data_as(C.POINTER(C.c_double))) if False else None
            # the env layer's ctrl mapping through the oracle itself: one env_step, recording every substep
            before_states = []
            # replicate ko_env_step: ctrl constant over the 15 substeps (oracle view after the first substep holds it)
            obs, rew, done, info = None, None, None, None
            st0 = (s.view("qpos").copy(), s.view("qvel").copy(), s.view("qacc_warmstart").copy())
            s.env_step(a, frame_skip=1); ctrl = s.view("ctrl").copy()
            s.set_state(*st0)
            for k in range(15):
                before = (s.view("qpos").copy(), s.view("qvel").copy(), s.view("qacc_warmstart").copy())
                s.step(ctrl)
                after = s.view("qpos").copy()
                qp, qv, qw, nc, con, st = lane.substep(*before, ctrl, hq[:, e])
                hull_contacts = sum(1 for c in s.contacts() if c["geom1"] != 0)
                if hull_contacts:
                    touching += 1
                    errs.append(np.abs(qp - after).max()); ncm += nc != s.s.ncon
    errs = np.array(errs)
    print(f"obj_first={obj_first}: {touching} substeps with a hull contact; fp32 lane one-step |dqpos| vs oracle: median {np.median(errs):.2e} p90 {np.percentile(errs,90):.2e} p99 {np.percentile(errs,99):.2e} max {errs.max():.2e}; "
          f"share > 1e-6: {(errs>1e-6).mean():.4f}  > 1e-5: {(errs>1e-5).mean():.4f}  > 1e-4: {(errs>1e-4).mean():.4f}; ncon mismatches {ncm}")
