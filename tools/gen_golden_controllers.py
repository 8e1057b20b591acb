#!/usr/bin/env python3
"""Dev-only: generate tests/golden/controllers.npz by calling the REFERENCE's scripted demonstrators
(expert_data.get_action with ExpertPIDController / NaiveController, expert_data.py:318-671) on random inputs.

Only the input/output vectors are committed; the reference is read from /root/reference by absolute path
(same stub modules as tests/golden/gen_golden_env.py).  The controllers look at obs[21] (object x in the palm frame),
obs[78], obs[79] (distal finger / object alignment), obs[81] (object / palm alignment), the controller's
initial obs[21], obs[81] and the lift flag.
"""
import contextlib
import io
import os
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np

REPO = Path(__file__).resolve().parents[1]
REF = Path("/root/reference/gym-kinova-gripper")
sys.path.insert(0, str(REPO / "tests" / "golden"))
from gen_golden_env import install_stubs  # noqa: E402


def main():
    install_stubs()
    os.chdir(REF)
    sys.path.insert(0, str(REF))
    sys.path.insert(0, str(REF / "gym_kinova_gripper" / "envs"))
    import expert_data
    rng = np.random.Generator(np.random.PCG64(20261002))
    n = 600
    obs = np.zeros((n, 82))
    init = np.zeros((n, 82))
    # object x across all regions of the controllers (centre, interpolation bands, extremes) incl. the band edges
    xs = rng.uniform(-0.07, 0.07, n)
    xs[:12] = [-0.04, -0.02, 0.02, 0.04, -0.03, 0.03, 0.0, -0.041, 0.041, -0.019, 0.019, 0.0301]
    obs[:, 21] = xs
    init[:, 21] = np.where(rng.random(n) < 0.7, xs, rng.uniform(-0.07, 0.07, n))     # the object may have moved since the start
    init[:12, 21] = xs[:12]
    init[:, 81] = rng.uniform(0.2, 1.0, n)
    d = rng.choice([0.0, 0.004, 0.02, 0.2], n) * rng.choice([-1, 1], n)
    obs[:, 81] = np.clip(init[:, 81] + d, 0.0, 1.0)
    obs[rng.random(n) < 0.15, 81] = rng.uniform(0.992, 1.0, int((rng.random(n) < 0.15).sum()) or 1)[0]
    obs[:, 78] = rng.uniform(0.0, 1.0, n)
    obs[:, 79] = rng.uniform(0.0, 1.0, n)
    lift = rng.random(n) < 0.4
    env = SimpleNamespace(action_space=SimpleNamespace(low=-0.8, high=0.8))
    out = {}
    for mode in ("naive", "position-dependent", "combined"):
        acts = np.zeros((n, 4))
        for i in range(n):
            with contextlib.redirect_stdout(io.StringIO()):
                ctl = expert_data.ExpertPIDController(init[i])
                acts[i] = expert_data.get_action(obs[i].copy(), bool(lift[i]), ctl, env, pid_mode=mode)
        out["action_" + mode.replace("-", "_")] = acts
    dst = REPO / "tests" / "golden" / "controllers.npz"
    np.savez_compressed(dst, obs21=obs[:, 21], obs78=obs[:, 78], obs79=obs[:, 79], obs81=obs[:, 81], init21=init[:, 21], init81=init[:, 81],
                        lift=lift, **out)
    print("wrote", dst, {k: v.shape for k, v in out.items()})
    for k, v in out.items():
        print(k, "distinct finger values:", np.unique(np.round(v[:, 1:], 6)).size)


if __name__ == "__main__":
    main()
