#!/usr/bin/env python3
"""Study + test helper (needs a GPU; test infrastructure - the oracle is the checker): long-horizon parity of the PRODUCT's env-step path.

tests/studies/long_horizon.py drives the GPU through ks_substep - one mj_step per call, every narrow-phase query COLD, as the oracle's.  The product
steps through ks_step / ks_rollout: 15 substeps per call with the lanes' pair memory carried from substep to substep and from call to call
(ks_core.h: PairWarm).  This study runs the same 168 grasp-and-lift envs (14 shapes x 3 poses x 4 starts, closing grasp + lift script) and the
config-2 random-action batch through ks_step and compares qpos with the oracle's env_step after every env-step (15, 30, ... substeps).
usage (GPU box): python -m tests.studies.long_horizon_envstep [n_env_steps]"""
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from kinovagrasping_amd import scenarios  # noqa: E402
from kinovagrasping_amd.sim import SOLVER_ITERATIONS, KinovaSim  # noqa: E402
from oracle import ko_py as ko  # noqa: E402

TOL = 1e-4


def run_batch(shape, q0, hq, actions, workers=32, precision=32):
    """q0 [16, n], hq [4, n], actions [T, 4, n].  Returns rel [T, n]: relative qpos error after every env-step."""
    n, T = q0.shape[1], actions.shape[0]
    model = ko.OracleModel(scenarios.model_blob(shape))
    orc = [ko.OracleSim(model, hq[:, i].copy(), solver_iterations=SOLVER_ITERATIONS) for i in range(n)]
    for i, o in enumerate(orc):
        o.s.rays_enabled = 0
        o.env_reset(q0[:, i].copy())
    sim = KinovaSim(n, shape, solver_iterations=SOLVER_ITERATIONS, horizon=0, precision=precision)
    sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
    rel = np.zeros((T, n))
    pool = ThreadPoolExecutor(workers)
    for t in range(T):
        a = actions[t]
        sim.step(torch.as_tensor(a))

        def ostep(i):
            orc[i].env_step(a[:, i])
            return orc[i].view("qpos").copy()
        qo = np.stack(list(pool.map(ostep, range(n))), 1)
        qg = sim.get_state()["qpos"].double().cpu().numpy()
        rel[t] = np.abs(qg - qo).max(0) / np.maximum(1e-3, np.abs(qo).max(0))
    status = sim.get_state()["status"].cpu().numpy()
    sim.close()
    pool.shutdown()
    return dict(rel=rel, status=status)


def shapes_batches(per, T, shapes=None, precision=32):
    out = {}
    script = np.array([[0.0, 0.6, 0.5, 0.7]] * 14 + [[0.6, 0.5, 0.5, 0.5]] * max(0, T - 14))[:T]
    for sh in (shapes or scenarios.SHAPES):
        qs, hqs = [], []
        for o in ("normal", "rotated", "top"):
            tab = scenarios.start_coord_table(sh, o)
            for r in np.linspace(0, len(tab) - 1, per).astype(int):
                q = np.zeros(16)
                q[9:12], q[12] = tab[r], 1.0
                q[0:3] = scenarios.hand_slide_offsets(o, sh, "pose")
                qs.append(q); hqs.append(scenarios.hand_quat_for(o))
        q0, hq = np.stack(qs, 1), np.stack(hqs, 1)
        out[sh] = run_batch(sh, q0, hq, np.repeat(script[:, :, None], q0.shape[1], 2), precision=precision)
    return out


def config2_batch(n, T):
    q0, hq = scenarios.config2_states(n)
    return run_batch("CubeS", q0, hq, scenarios.config_actions(n, T))


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 14          # 14 env-steps = 210 substeps
    r = config2_batch(256, T)
    k = min(T, 14) - 1
    print(f"config 2 x 256 through ks_step: within 1e-4 after env-step {k + 1} ({15 * (k + 1)} substeps): {np.mean(r['rel'][k] <= TOL):.3f}; never beyond on the way: "
          f"{np.mean((r['rel'][:k + 1] <= TOL).all(0)):.3f}; median {np.median(r['rel'][k]):.1e}")
    res = shapes_batches(4, T)
    within = {sh: int((x["rel"][k] <= TOL).sum()) for sh, x in res.items()}
    within13 = {sh: int((x["rel"][k - 1] <= TOL).sum()) for sh, x in res.items()}
    print(f"grasp-and-lift scripts through ks_step, envs of 12 within 1e-4 after env-step {k + 1} ({15 * (k + 1)} substeps):", within, "total", sum(within.values()), "of", 12 * len(within))
    print(f"   after env-step {k} ({15 * k} substeps): total", sum(within13.values()), "; median rel per shape at the end:", {sh: f"{np.median(x['rel'][k]):.1e}" for sh, x in res.items()})


if __name__ == "__main__":
    main()
