#!/usr/bin/env python3
"""Study (CPU; test infrastructure - the oracle is the checker): WHAT decides the first divergence of the fp32 product from the fp64 oracle in the
grasp-and-lift long-horizon runs ("a table showing which quantity's rounding decides the first divergence ... per env: first
differing contact toggle, facet jump or rim tie").

The fp32 kernel source runs on the host (tests/native_build.Lane: one lane of ks_core.h, cold queries), free-running for 200 substeps from the 168
starts of tests/studies/long_horizon.shapes_batches (14 shapes x 3 poses x 4 starts, closing grasp + lift script), beside the fp64 oracle.  At every
substep BOTH step from the fp32 lane's state (teacher forcing): the one-step outcome differs by fp32 rounding (~1e-7) unless a discrete decision fell
differently, and the FIRST such event of an env is classified from the two contact lists:
    toggle   a pair touches on one side only (contact count / pair set differs): depth within rounding of 0 (or of the margin)
    facet    same pairs, a contact NORMAL differs by > 1e-3 rad: the penetration / distance query ended on another facet (hull pair) -
             or, for the ground plane, never (its normal is fixed)
    point    same pairs and normals, a contact POINT of a hull pair differs by > 1e-5 m: the portal ended elsewhere on the same facet
    rim      the contact SET of a plane pair differs (compared as a set: the order of a plane pair's contacts is of no consequence): other vertices of
             the hull's rim / base on the floor
    solver   same contacts (points within 1e-5, normals within 1e-3), one-step qpos still differs by > 20 x the typical rounding: Newton's active set
and by the pair it happened on (on the fp32 lane with round 4's arithmetic, -DKS_REFINE_F64=0 -DKS_PLANE_F64=0, on round 5's float32 hull tables).  Then: the same runs on host builds with one thing
changed at a time (VARIANTS); on the fp64 lane with Gaussian noise added to qpos / qvel after every substep (how much noise the discontinuities tolerate: the
fp32 state's own rounding is ~3e-8 relative); and on a MIXED lane (tests/native/ks_lanecheck.cpp: substep_mixed) with fp64 in one stage at a time - which stage's
precision decides.  Outcome (round 5): not the state and not the solver - quantities of the collision stage that keep the same rounding error while a contact rests on
the same vertices: the depth read off MPR's final portal and the plane pairs' vertex distances (which also pick the rim vertices of a round base).
usage: python -m tests.studies.divergence_table > profiles/r05_divergence_table.txt"""
import sys
from collections import Counter
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from kinovagrasping_amd import scenarios  # noqa: E402
from kinovagrasping_amd.sim import SOLVER_ITERATIONS  # noqa: E402

GN = ["ground", "palm", "f1_prox", "f1_dist", "f2_prox", "f2_dist", "f3_prox", "f3_dist", "object"]
N_SUB, PER, TOL = 200, 4, 1e-4


def starts(shape):
    T = (N_SUB + 14) // 15
    script = np.array([[0.0, 0.6, 0.5, 0.7]] * 14 + [[0.6, 0.5, 0.5, 0.5]] * max(0, T - 14))[:T]
    out = []
    for o in ("normal", "rotated", "top"):
        tab = scenarios.start_coord_table(shape, o)
        for r in np.linspace(0, len(tab) - 1, PER).astype(int):
            q = np.zeros(16)
            q[9:12], q[12] = tab[r], 1.0
            q[0:3] = scenarios.hand_slide_offsets(o, shape, "pose")
            out.append((o, q, scenarios.hand_quat_for(o), script))
    return out


def classify(con_l, nc_l, cons_o):
    """first difference between the lane's contact records and the oracle's contact list.  The (up to four) contacts of a plane pair are compared as
    a SET: their order follows which vertex is deepest and is of no consequence."""
    if nc_l != len(cons_o):
        return "toggle", "count"
    k = 0
    while k < len(cons_o):
        c = cons_o[k]
        name = f"{GN[min(c['geom1'], 8)]}-{GN[min(c['geom2'], 8)]}"
        if c["geom1"] == 0:                                   # plane pair: gather its run of contacts
            run = [j for j in range(k, len(cons_o)) if cons_o[j]["geom1"] == 0 and cons_o[j]["geom2"] == c["geom2"]]
            P_o = np.array([cons_o[j]["pos"] for j in run]); P_l = np.array([con_l[j][:3] for j in run])
            d = np.abs(P_o[:, None, :] - P_l[None, :, :]).max(2)
            if d.min(1).max() > 1e-5 or d.min(0).max() > 1e-5:
                return "rim", name                            # another VERTEX SET of the hull on the floor
            k = run[-1] + 1
            continue
        n_l, p_l = con_l[k][3:6], con_l[k][:3]
        ang = np.arccos(np.clip(float(n_l @ c["frame"][:3]), -1, 1))
        if ang > 1e-3:
            return "facet", name
        if np.abs(p_l - c["pos"]).max() > 1e-5:
            return "point", name
        k += 1
    return None, None


VARIANTS = {   # host builds of the kernel source with experiment switches (fp32 lane); built once by the parent process into /tmp
    "r4": ["-DKS_MPR_SM=0", "-DKS_REFINE_F64=0", "-DKS_PLANE_F64=0"],                                  # round 4's fp32 read-offs
    "r4+planehook": ["-DKS_MPR_SM=0", "-DKS_REFINE_F64=0", "-DKS_PLANE_F64=0", "-DKS_PLANE_HOOK"],     # ... with every plane pair's first vertex from an fp64 evaluation
    "r4+mink64": ["-DKS_MPR_SM=0", "-DKS_REFINE_F64=0", "-DKS_PLANE_F64=0", "-DKS_MINK_F64=1"],        # ... with the Minkowski points formed in fp64
    "r5a": ["-DKS_MPR_SM=0", "-DKS_PLANE_F64=0"],                                  # depth / direction of MPR's final portal in fp64 (KS_REFINE_F64=1), plane pairs fp32
    "r5a+plane1": ["-DKS_MPR_SM=0", "-DKS_PLANE_F64=1"],                           # ... + the staged plane contacts' depths in fp64
    "r5": ["-DKS_MPR_SM=0"],                                      # round 5's product arithmetic, cold queries (the lane's; the GPU's queries were warm: "r5warm/warm")
    "r5warm": ["-DKS_MPR_SM=0", "-DKS_MPR_WARM=1"],               # ... as the GPU ran it through ks_step (run as "r5warm/warm": the lane keeps its pair memory)
    "r6": [],
    "r6first": ["-DKS_MPR_FIRST=3"],                              # ... and a pair that penetrated in the previous substep skips the distance query                                                     # round 6, the product: the penetration query cold, on fp64 Minkowski points, one support site
}


def _variant_so(name):
    return f"/tmp/libks_lanecheck_{name.replace('+', '_')}.so"


def build_variants(names=None):
    import subprocess
    from tests import native_build
    for name, flags in VARIANTS.items():
        if names is not None and name not in names:
            continue
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared"] + flags + ["-o", _variant_so(name), str(native_build.HERE / "ks_lanecheck.cpp")])
    native_build.lanecheck_lib(); native_build.lanecheck_lib(True)


def _variant_lib(name):
    def load(multi_geom=False):
        import ctypes as C
        from tests import native_build
        L = C.CDLL(_variant_so(name))
        L.lc_create.restype = C.c_void_p; L.lc_create.argtypes = [C.c_char_p, C.c_size_t]
        L.lc_substep.argtypes = [C.c_void_p, C.c_int, native_build.dp, native_build.dp, native_build.dp, native_build.dp, native_build.dp, C.c_int, C.POINTER(C.c_int), native_build.dp]
        return L
    return load


def run_env(args):
    shape, pose, q0, hq, script, mode, eps, seed = args
    from oracle import ko_py as ko
    from tests import native_build
    warm_queries = mode.startswith("fp32w:")         # the lane keeps its pair memory from substep to substep, as the GPU's lanes do
    if mode.startswith("fp32:") or warm_queries:
        native_build.lanecheck_lib = _variant_lib(mode.split(":", 1)[1])
        mode = "fp32"
    blob = scenarios.model_blob(shape)
    m = ko.OracleModel(blob)
    ref = ko.OracleSim(m, hq, solver_iterations=SOLVER_ITERATIONS); ref.s.rays_enabled = 0
    ref.env_reset(q0.copy())
    tf = ko.OracleSim(m, hq, solver_iterations=SOLVER_ITERATIONS); tf.s.rays_enabled = 0          # teacher-forced twin of the lane
    lane = native_build.Lane(blob, {"fp32": 32, "fp64": 64}.get(mode, 6432))           # 6432: ks_lanecheck.cpp substep_mixed
    if mode.startswith("mixed"):
        lane.L.lc_set_mixed_variant(int(mode[5:] or 0))
    if warm_queries:
        lane.L.lc_set_warm.argtypes = [__import__("ctypes").c_void_p, __import__("ctypes").c_int]
        lane.L.lc_set_warm(lane.h, 1)
    rng = np.random.default_rng(seed)
    st = (ref.view("qpos").copy(), ref.view("qvel").copy(), ref.view("qacc_warmstart").copy())
    first_event, rel_end, first_bad, base = None, 0.0, -1, []
    ctrl = np.zeros(9)
    for k in range(N_SUB):
        if k % 15 == 0:
            ctrl = ko.env_ctrl(ref.view("geom_xpos").reshape(-1, 3)[1], ref.view("geom_xmat").reshape(-1, 9)[1], script[k // 15])[2]
        ref.step(ctrl)
        qp, qv, qw, nc, con, status = lane.substep(*st, ctrl, hq)
        if mode == "fp32" and first_event is None:
            tf.set_state(*st); tf.step(ctrl)
            d = np.abs(qp - tf.view("qpos")).max()
            kind, where = classify(con, nc, tf.contacts())
            typical = np.median(base) if len(base) >= 5 else 1e-7
            if kind is None and d > max(20 * typical, 2e-6):
                kind, where = "solver", "-"
            if kind is not None:
                depth = min([c["dist"] for c in tf.contacts()] + [1.0])
                first_event = (k, kind, where, d)
            else:
                base.append(d)
        if eps > 0:
            qp = qp.copy(); qp[:12] += eps * np.maximum(1e-3, np.abs(qp[:12])) * rng.standard_normal(12)
        elif eps < 0:                                  # absolute Gaussian noise of |eps| (m/s, rad/s) on qvel: what a noisy ACCELERATION leaves behind
            qv = qv.copy(); qv += -eps * rng.standard_normal(15)
        st = (qp, qv, qw)
        qo = ref.view("qpos")
        rel = np.abs(qp - qo).max() / max(1e-3, np.abs(qo).max())
        if rel > TOL and first_bad < 0:
            first_bad = k
        rel_end = rel
    return shape, pose, first_event, first_bad, rel_end


def run_variant(name):
    jobs = []
    warm = name.endswith("/warm")
    name = name[:-5] if warm else name
    for sh in scenarios.SHAPES:
        for i, (o, q, hq, script) in enumerate(starts(sh)):
            jobs.append((sh, o, q, hq, script, f"{'fp32w' if warm else 'fp32'}:{name}", 0.0, i))
    with ProcessPoolExecutor(8) as ex:
        return list(ex.map(run_env, jobs, chunksize=4))


def variants_only(names):
    build_variants([n[:-5] if n.endswith("/warm") else n for n in names])
    for name in names:
        rv = run_variant(name)
        per = {}
        for x in rv:
            per[x[0]] = per.get(x[0], 0) + int(x[4] <= TOL)
        print(f"   {name:14s} {sum(r[4] <= TOL for r in rv):3d}   median rel at 200 {np.median([r[4] for r in rv]):.1e}   {dict(Counter(r[2][1] if r[2] else 'none' for r in rv if r[4] > TOL))}   per shape {per}", flush=True)


def main():
    build_variants()

    res = run_variant("r4")
    ok = sum(r[4] <= TOL for r in res)
    print(f"fp32 kernel lane WITH ROUND 4's ARITHMETIC (host build of ks_core.h with -DKS_REFINE_F64=0 -DKS_PLANE_F64=0) vs fp64 oracle, free running, {len(res)} envs x {N_SUB} substeps: {ok} within 1e-4 at substep {N_SUB}")
    print("first discrete event per env under teacher forcing (the one-step outcomes of lane and oracle from the lane's own state):\n")
    print(f"{'shape':10s} {'pose':8s} {'first event':>11s} {'kind':>7s} {'pair':>18s} {'one-step |dq|':>13s} {'first > 1e-4':>12s} {'rel at 200':>10s}")
    for sh, o, ev, fb, rel in res:
        k, kind, where, d = ev if ev else (-1, "-", "-", 0.0)
        print(f"{sh:10s} {o:8s} {k:11d} {kind:>7s} {where:>18s} {d:13.2e} {fb:12d} {rel:10.1e}")
    bad = [r for r in res if r[4] > TOL]
    good = [r for r in res if r[4] <= TOL]
    print(f"\nenvs beyond 1e-4 at substep {N_SUB}: {len(bad)}.  Their first event by kind: {dict(Counter(r[2][1] if r[2] else 'none' for r in bad))}")
    print(f"   by pair: {dict(Counter((r[2][1] + ' ' + r[2][2]) if r[2] else 'none' for r in bad).most_common())}")
    lead = [r[3] - r[2][0] for r in bad if r[2] and r[3] >= 0]
    print(f"   substeps from the first event to the first relative error beyond 1e-4: median {np.median(lead):.0f}, p10 {np.percentile(lead, 10):.0f}, p90 {np.percentile(lead, 90):.0f}"
          f"; envs whose error crossed 1e-4 BEFORE any event: {sum(1 for x in lead if x < 0)}")
    print(f"envs within 1e-4: {len(good)}; of them with an event on the way: {sum(1 for r in good if r[2])} (an event need not separate the trajectories for good: "
          f"{dict(Counter(r[2][1] for r in good if r[2]))})")
    # ---- is the most frequent first event the cause?  And what is?  The same runs on builds with one thing changed at a time
    print("\nthe same 168 runs on host builds of the fp32 lane with ONE thing changed (envs within 1e-4 at substep 200; first events of the envs beyond it):")
    counts = {}
    for name, what in (("r4+planehook", "round 4 + every plane pair's FIRST VERTEX from an fp64 evaluation on the lane's own pose (host-only hook: no 'rim' event can occur)"),
                       ("r4+mink64", "round 4 + the Minkowski points of the support pairs formed in fp64 (KS_MINK_F64=1)"),
                       ("r5a", "depth and direction of MPR's FINAL portal recomputed in fp64 from its vertex ids (KS_REFINE_F64=1)"),
                       ("r5a+plane1", "... + the staged plane contacts' depths formed in fp64 (KS_PLANE_F64=1)"),
                       ("r5", "ROUND 5, THE PRODUCT: ... + the plane pairs' vertex scans in fp64: which vertex is deepest, which are within the margin (KS_PLANE_F64=2)")):
        rv = run_variant(name)
        counts[name] = sum(r[4] <= TOL for r in rv)
        print(f"   {name:14s} {counts[name]:3d}   median rel at 200 {np.median([r[4] for r in rv]):.1e}   {dict(Counter(r[2][1] if r[2] else 'none' for r in rv if r[4] > TOL))}   <- {what}", flush=True)
    print(f"   -> taking the most frequent FIRST difference away (plane hook: {counts['r4+planehook']} against {ok}) is not what brings the trajectories together.  What does are quantities that keep\n"
          f"      the SAME rounding error for as long as a contact rests on the same vertices - a bias, not noise: the depth read off MPR's final portal ({counts['r5a']}), and the plane\n"
          f"      pairs' vertex distances, which also decide WHICH of a round base's almost equally deep rim vertices carry the contacts ({counts['r5']}; the model's hull tables are\n"
          f"      float32 numbers in the oracle and the kernels alike since round 5).  An all-fp64 collision stage on the fp32 poses: see MIXED below.")
    # ---- how much state noise the discontinuities tolerate
    print("\nfp64 lane with relative Gaussian noise eps on qpos[0:12] after every substep (same 168 runs): envs within 1e-4 at substep 200")
    for eps in (0.0, 1e-9, 3e-8, 1e-7):
        jobs = []
        for sh in scenarios.SHAPES:
            for i, (o, q, hq, script) in enumerate(starts(sh)):
                jobs.append((sh, o, q, hq, script, "fp64", eps, 1000 + i))
        with ProcessPoolExecutor(8) as ex:
            r2 = list(ex.map(run_env, jobs, chunksize=4))
        print(f"   eps {eps:7.0e}: {sum(r[4] <= TOL for r in r2):3d} of {len(r2)}   (median rel at 200: {np.median([r[4] for r in r2]):.1e})", flush=True)


def mixed_precision(only=None):
    """VERDICT r4 next #4, measured on the host before anything is built for the GPU: kinematics, mass matrix, smooth forces and the WHOLE collision stage in
    fp64 with the state kept in fp64, the constraint solver + Euler step in fp32 on the rounded scratch (tests/native/ks_lanecheck.cpp: substep_mixed)."""
    print("\nMIXED precision on the host lane (tests/native/ks_lanecheck.cpp: substep_mixed; the solver + Euler step always fp32; state kept in fp64 between substeps):")
    for variant, what in ((1, "fp64 kinematics + mass matrix + smooth forces (state kept in fp64), collision = the product's fp32 stage (with its fp64 read-offs) on the rounded poses"),
                          (32, "fp32 state and kinematics, the whole COLLISION stage fp64 on those poses"),
                          (33, "fp32 state and kinematics, HULL pairs (GJK / MPR) from an fp64 collision on those poses, plane pairs the product's"),
                          (34, "fp32 state and kinematics, PLANE pairs from an fp64 collision on those poses, hull pairs the product's (~ the product, 146 on the fp32 lane: rows that differ by < 8 differ by the noise of this count)"),
                          (2, "as 32 with qpos / qvel accumulated in fp64 over the substeps"),
                          (5, "the product's fp32 stages, qpos AND qvel accumulated in fp64 over the substeps (the one-step increments are the fp32 product's)"),
                          (6, "... qpos accumulated in fp64, qvel rounded to fp32 after every substep"),
                          (7, "... qvel accumulated in fp64, qpos rounded to fp32 after every substep")):
        if only is not None and variant not in only:
            continue
        jobs = []
        for sh in scenarios.SHAPES:
            for i, (o, q, hq, script) in enumerate(starts(sh)):
                jobs.append((sh, o, q, hq, script, f"mixed{variant}", 0.0, i))
        with ProcessPoolExecutor(8) as ex:
            r = list(ex.map(run_env, jobs, chunksize=4))
        per = {}
        for x in r:
            per[x[0]] = per.get(x[0], 0) + int(x[4] <= TOL)
        print(f"   {what}: {sum(x[4] <= TOL for x in r)} of {len(r)} within 1e-4 at substep {N_SUB} (median rel {np.median([x[4] for x in r]):.1e}); per shape {per}", flush=True)


def velocity_noise():
    print("\nfp64 lane with absolute Gaussian noise sigma on qvel after every substep (a noisy solver: the fp32 lane's own one-step |dqvel| against the oracle\n"
          "from equal states is median 1 - 2.5e-6, p90 3 - 7e-6, p99 1.7e-5 m/s - the fp32 rounding of contact forces of 10 - 25 N that cancel on a 0.1 kg\n"
          "object): envs within 1e-4 at substep 200")
    for sig in (1e-7, 1e-6, 1e-5):
        jobs = []
        for sh in scenarios.SHAPES:
            for i, (o, q, hq, script) in enumerate(starts(sh)):
                jobs.append((sh, o, q, hq, script, "fp64", -sig, 2000 + i))
        with ProcessPoolExecutor(8) as ex:
            r2 = list(ex.map(run_env, jobs, chunksize=4))
        print(f"   sigma {sig:7.0e}: {sum(r[4] <= TOL for r in r2):3d} of {len(r2)}   (median rel at 200: {np.median([r[4] for r in r2]):.1e})", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "velocity":
        velocity_noise()
    elif len(sys.argv) > 1 and sys.argv[1] == "variants":
        variants_only(sys.argv[2:])
    elif len(sys.argv) > 1 and sys.argv[1] == "mixed":
        mixed_precision([int(a) for a in sys.argv[2:]] or None)
    else:
        main()
        velocity_noise()
        mixed_precision()
