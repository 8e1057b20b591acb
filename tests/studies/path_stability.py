#!/usr/bin/env python3
"""Study (CPU, host lane of the kernel source): how often does a pair's cold penetration query take the SAME sequence of support pairs as the pair's
query of the previous substep?  (The premise of a speculative, lane-parallel verification of the query: round 6)
168 grasp-and-lift envs (tests/studies/divergence_table.starts) x 200 substeps, fp32 lane free running.
usage: python -m tests.studies.path_stability"""
import ctypes as C
import subprocess
import sys
from collections import Counter
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from kinovagrasping_amd import scenarios  # noqa: E402
from tests.studies.divergence_table import starts, N_SUB  # noqa: E402

SO = "/tmp/libks_lanecheck_pathstudy.so"


def build():
    from tests import native_build
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DKS_PATH_STUDY", "-o", SO, str(native_build.HERE / "ks_lanecheck.cpp")])


def run_env(args):
    shape, pose, q0, hq, script = args
    from oracle import ko_py as ko
    from tests import native_build

    def load(multi_geom=False):
        L = C.CDLL(SO)
        L.lc_create.restype = C.c_void_p; L.lc_create.argtypes = [C.c_char_p, C.c_size_t]
        L.lc_substep.argtypes = [C.c_void_p, C.c_int, native_build.dp, native_build.dp, native_build.dp, native_build.dp, native_build.dp, C.c_int, C.POINTER(C.c_int), native_build.dp]
        return L
    native_build.lanecheck_lib = load
    blob = scenarios.model_blob(shape)
    m = ko.OracleModel(blob)
    ref = ko.OracleSim(m, hq, solver_iterations=20); ref.s.rays_enabled = 0
    ref.env_reset(q0.copy())
    lane = native_build.Lane(blob, 32)
    lane.L.lc_path_reset()
    st = (ref.view("qpos").copy(), ref.view("qvel").copy(), ref.view("qacc_warmstart").copy())
    ctrl = np.zeros(9)
    for k in range(N_SUB):
        if k % 15 == 0:
            ctrl = ko.env_ctrl(ref.view("geom_xpos").reshape(-1, 3)[1], ref.view("geom_xmat").reshape(-1, 9)[1], script[k // 15])[2]
        ref.step(ctrl)
        qp, qv, qw, nc, con, status = lane.substep(*st, ctrl, hq)
        st = (qp, qv, qw)
        lane.L.lc_path_tick()
    buf = (C.c_int * 400000)()
    n = lane.L.lc_path_log(buf, 400000)
    return shape, pose, np.array(buf[:n]).reshape(-1, 5)


def main():
    build()
    jobs = [(sh, o, q, hq, script) for sh in scenarios.SHAPES for (o, q, hq, script) in starts(sh)]
    with ProcessPoolExecutor(8) as ex:
        res = list(ex.map(run_env, jobs, chunksize=4))
    A = np.concatenate([r[2] for r in res])
    gap, nprev, ncur, k, tick = A.T
    print(f"queries {len(A)}; turns per query (incl. the terminal record): mean {ncur.mean() - 1:.2f}, histogram {dict(sorted(Counter((ncur - 1).tolist()).items()))}")
    cont = gap == 1
    print(f"queries whose pair ran a query in the previous substep: {cont.sum()} ({cont.mean():.3f})")
    same = cont & (k == ncur) & (nprev == ncur)
    print(f"   identical path and outcome: {same.sum()} = {same.sum() / cont.sum():.3f} of them, {same.sum() / len(A):.3f} of all")
    rem = np.where(cont, ncur - 1 - np.minimum(k, ncur - 1), ncur - 1)
    print(f"   serial turns left after the verified prefix (all queries, no memory = all turns): mean {rem.mean():.2f} of {(ncur - 1).mean():.2f}")
    d = cont & ~same
    print(f"   of the {d.sum()} that differ: first differing turn histogram {dict(sorted(Counter(k[d].tolist()).items()))}")
    # per substep and env: does EVERY query of the env verify?  (the wave pays for its slowest lane)
    per_env = []
    for sh, o, a in res:
        if len(a) == 0:
            continue
        ok = (a[:, 0] == 1) & (a[:, 3] == a[:, 2]) & (a[:, 1] == a[:, 2])
        ticks = np.unique(a[:, 4])
        all_ok = np.array([ok[a[:, 4] == t].all() for t in ticks])
        per_env.append((sh, o, len(ticks), all_ok.mean(), ok.mean()))
    tot = sum(p[2] for p in per_env)
    print(f"env-substeps with >= 1 query: {tot}; with ALL their queries verified: {sum(p[2] * p[3] for p in per_env) / tot:.3f}")
    # by episode phase (substep index)
    for lo in range(0, N_SUB, 25):
        sel = (tick >= lo) & (tick < lo + 25)
        if sel.sum():
            print(f"   substeps {lo:3d}-{lo + 24:3d}: queries {sel.sum():5d}, previous-substep query {cont[sel].mean():.3f}, identical {same[sel].sum() / max(1, sel.sum()):.3f}, serial turns left {rem[sel].mean():.2f} of {(ncur[sel] - 1).mean():.2f}")


if __name__ == "__main__":
    main()
