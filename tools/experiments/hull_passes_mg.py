#!/usr/bin/env python3
"""Dev tool (GPU box; -DKS_STAMP -DKS_MULTI_GEOM build as tools/experiments/build/libkinova_sim_mg_stamp.so, KS_LIB_MG pointing at it): narrow-phase passes per
wave and substep of the multi-geom library under tools/debug/mg_perf.py's protocol (random actions, auto-reset).  usage: hull_passes_mg.py BowlS [closing]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from kinovagrasping_amd import scenarios, model_compiler as mc
from kinovagrasping_amd.sim import KinovaSim
n = 4096
shape = sys.argv[1] if len(sys.argv) > 1 else "BowlS"
mode = sys.argv[2] if len(sys.argv) > 2 else "random"
M = mc.read_blob(scenarios.model_blob(shape))
g = M["geom_pos"][8]
rng = np.random.default_rng(0)
q0 = np.zeros((16, n)); q0[12] = 1
q0[9] = -g[0] + rng.uniform(-0.03, 0.03, n); q0[10] = -g[1] + rng.uniform(-0.01, 0.01, n); q0[11] = 0.0
hq = np.repeat(scenarios.hand_quat_for("normal")[:, None], n, 1)
sim = KinovaSim(n, shape, auto_reset=True, horizon=30, contact_tap=True)
sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
base = scenarios.config_actions(256, 30)
acts = torch.as_tensor(np.tile(base, (1, 1, n // 256))).cuda()
closing = torch.tensor([0.0, 0.6, 0.5, 0.7], device='cuda').repeat(n, 1).t().contiguous()
tot = np.zeros(3); cnt = 0
ncon_max = sim.ncon_max if hasattr(sim, "ncon_max") else 40
for t in range(60):
    sim.step(closing if mode == "closing" else acts[t % 30])
    if t < 30:
        continue
    st = sim.get_state(contacts=True)
    torch.cuda.synchronize()
    prof = st["contact"].reshape(-1, n)[:480].cpu().numpy().reshape(16, 30, n)
    lane_pairs, wave_passes = prof[:, 24], prof[:, 25]
    tot += [lane_pairs.sum(0).mean() / 15, wave_passes[0].mean() / 15, lane_pairs.max(0).mean() / 15]; cnt += 1
print(f"{shape} ({mode} actions), 30 env-steps x 15 substeps x {n} envs: live hull pairs per env and substep {tot[0] / cnt:.2f}; narrow-phase passes per wave and substep "
      f"{tot[1] / cnt:.3f}; an env's live pairs spread over its 16 lanes would need {np.ceil(tot[0] / cnt / 16):.0f}")
