#!/usr/bin/env python3
"""Round 6 A/B (GPU): are two builds of the library bit-identical per env?  168 grasp-and-lift envs (14 shapes x 3 poses x 4 starts, closing grasp + lift
script) through ks_step, qpos / qvel dumped after every env-step; run once per build (KS_LIB=...), then `ab_bits.py cmp a.npz b.npz`.
usage: KS_LIB=... python tools/r06/ab_bits.py run out.npz [n_env_steps] | python tools/r06/ab_bits.py cmp a.npz b.npz"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))


def run(out, T):
    import torch
    from kinovagrasping_amd import scenarios
    from kinovagrasping_amd.sim import SOLVER_ITERATIONS, KinovaSim
    script = np.array([[0.0, 0.6, 0.5, 0.7]] * 14 + [[0.6, 0.5, 0.5, 0.5]] * max(0, T - 14))[:T]
    res = {}
    for sh in scenarios.SHAPES:
        qs, hqs = [], []
        for o in ("normal", "rotated", "top"):
            tab = scenarios.start_coord_table(sh, o)
            for r in np.linspace(0, len(tab) - 1, 4).astype(int):
                q = np.zeros(16); q[9:12], q[12] = tab[r], 1.0
                q[0:3] = scenarios.hand_slide_offsets(o, sh, "pose")
                qs.append(q); hqs.append(scenarios.hand_quat_for(o))
        q0, hq = np.stack(qs, 1), np.stack(hqs, 1)
        n = q0.shape[1]
        sim = KinovaSim(n, sh, solver_iterations=SOLVER_ITERATIONS, horizon=0, precision=32, contact_tap=True)
        sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
        Q = []
        for t in range(T):
            sim.step(torch.as_tensor(np.repeat(script[t][:, None], n, 1)))
            st = sim.get_state(contacts=True)
            Q.append(np.concatenate([st["qpos"].cpu().numpy(), st["qvel"].cpu().numpy()], 0))
        res[sh] = np.stack(Q)
        sim.close()
    np.savez(out, **res)


def cmp(a, b):
    A, B = np.load(a), np.load(b)
    tot = same = 0
    for sh in A.files:
        x, y = A[sh], B[sh]                      # [T, 31, n]
        d = (x.view(np.uint32) != y.view(np.uint32)).any(1)      # [T, n]
        first = np.where(d.any(0), d.argmax(0), -1)
        tot += d.shape[1]; same += int((first < 0).sum())
        mx = [float(np.abs(x[f, :, i] - y[f, :, i]).max()) if f >= 0 else 0.0 for i, f in enumerate(first.tolist())]
        print(f"{'':10s} largest |difference| of qpos / qvel at that env-step: max {max(mx):.2e}, median {np.median([m for m in mx if m > 0] or [0]):.2e}")
        print(f"{sh:10s} envs {d.shape[1]:3d} bit-identical to the end {int((first < 0).sum()):3d}; first differing env-step per env {first.tolist()}")
    print(f"total {same} of {tot} envs bit-identical over {x.shape[0]} env-steps")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 14)
    else:
        cmp(sys.argv[2], sys.argv[3])
