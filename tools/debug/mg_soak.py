#!/usr/bin/env python3
"""Dev-only (GPU box): soak of the multi-geom library - 4096 envs x K env-steps per context, random actions, auto-reset, objects placed in
the hand; reports the sticky status flags (1 contact overflow, 2 non-finite, 4 ray time-out, 8 Newton cap) and the contact-count range."""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kinovagrasping_amd import model_compiler as mc, scenarios
from kinovagrasping_amd.sim import KinovaSim

n, K = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for shapes in (["BottleS"], ["BowlB"], ["TBottleM", "RBowlS", "BottleB", "BowlS", "CubeS", "Vase2B"]):
    sim = KinovaSim(n, shapes[0] if len(shapes) == 1 else shapes, auto_reset=True, horizon=30)
    oid = (np.arange(n) * len(shapes) // n).astype(np.int32)
    q = np.zeros((16, n)); q[12] = 1
    hq = np.zeros((4, n))
    rng = np.random.default_rng(0)
    for e in range(n):
        sh = shapes[oid[e]]
        pose = ("normal", "rotated", "top")[e % 3] if sh != "RBowlS" or e % 3 else "top"
        g = mc.read_blob(scenarios.model_blob(sh))["geom_pos"][8] if e == 0 or oid[e] != oid[e - 1] else g
        q[9:12, e] = (-g * np.array([1, 1, 0]) + np.append(rng.uniform(-0.04, 0.04, 2), 0.0)) if sh in scenarios.MULTI_GEOM_SHAPES else scenarios.start_coord_table(sh, pose)[e % 4000]
        q[0:3, e] = scenarios.hand_slide_offsets(pose, sh, "pose")
        hq[:, e] = scenarios.hand_quat_for(pose)
    sim.reset(torch.as_tensor(q), torch.as_tensor(hq), object_id=oid if len(shapes) > 1 else None)
    g_ = torch.Generator(device="cuda").manual_seed(1)
    t0 = time.time()
    ncmax = 0
    for t in range(K):
        sim.step(torch.rand((4, n), device="cuda", generator=g_) * 1.6 - 0.8)
        if t % 100 == 99:
            ncmax = max(ncmax, int(sim.get_state()["ncon"].max()))
    torch.cuda.synchronize()
    st = sim.get_state()
    status = st["status"].cpu().numpy()
    print(f"{'+'.join(shapes):50s} {n * K / 1e6:.1f} M env-steps in {time.time() - t0:.1f} s: status flags seen {sorted(set(status.tolist()))}, "
          f"envs with any flag {int((status != 0).sum())}, finite {bool(torch.isfinite(st['qpos']).all())}, max contacts sampled {ncmax}", flush=True)
    sim.close()
