#!/usr/bin/env python3
"""Dev-only (GPU box): stepping rate of a multi-geom context (libkinova_sim_mg.so) - ks_step with random actions, auto-reset -
against the standard library on a single-geom object.  KS_DEBUG=1 prints the launch plan (envs per workgroup, LDS)."""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kinovagrasping_amd import model_compiler as mc, scenarios
from kinovagrasping_amd.sim import KinovaSim

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for shapes in (["CubeS"], ["BottleS"], ["TBottleS"], ["BowlS"], ["RBowlS"], ["CubeS", "BottleS", "BowlS", "TBottleS"]):
    model = shapes[0] if len(shapes) == 1 else shapes
    sim = KinovaSim(n, model, auto_reset=True, horizon=30)
    q = np.zeros((16, n)); q[12] = 1
    oid = (np.arange(n) * len(shapes) // n).astype(np.int32)
    for e in range(n):
        sh = shapes[oid[e]]
        M = mc.read_blob(scenarios.model_blob(sh)) if e == 0 or oid[e] != oid[e - 1] else M
        q[9:12, e] = -M["geom_pos"][8] * np.array([1, 1, 0]) + np.array([0.03 * np.sin(e), 0.02 * np.cos(e), 0.0])
        if not sim.multi_geom or sh in scenarios.SHAPES:
            q[9:12, e] = scenarios.start_coord_table(sh)[e % 4000]
    hq = np.repeat(scenarios.hand_quat_for("normal")[:, None], n, 1)
    sim.reset(torch.as_tensor(q), torch.as_tensor(hq), object_id=oid if len(shapes) > 1 else None)
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = [torch.rand((4, n), device="cuda", generator=g) * 0.8 for _ in range(8)]
    for t in range(10):
        sim.step(acts[t % 8])
    torch.cuda.synchronize()
    t0 = time.time()
    K = 60
    for t in range(K):
        sim.step(acts[t % 8])
    torch.cuda.synchronize()
    dt = time.time() - t0
    st = sim.get_state()
    print(f"{'+'.join(shapes):32s} lib {'mg ' if sim.multi_geom else 'std'} {n} envs: {n * K / dt / 1e6:.3f} M env-steps/s ({dt / K * 1e3:.3f} ms per env-step), status bits {int(st['status'].max())}, mean contacts {float(st['ncon'].float().mean()):.2f}", flush=True)
    sim.close()
