#!/usr/bin/env python3
"""Dev tool (GPU box; diagnostic build -DKS_STAMP as tools/experiments/build/libkinova_sim_stamp.so, run with KS_LIB pointing at it): how many narrow-phase
PASSES does a wave run per substep in the bench's training regime?  A lane owns up to two hull pairs; the wave runs as many passes as its busiest lane has
live pairs.  Prints live pairs per env, passes per wave, and what a dealing that spreads an env's live pairs over its 16 lanes would need."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from kinovagrasping_amd import scenarios
from kinovagrasping_amd.sim import KinovaSim
from kinovagrasping_amd.ddpgfd import DDPGfD
from kinovagrasping_amd.rollout import RolloutEngine
from kinovagrasping_amd.replay import DeviceEpisodeReplay
from kinovagrasping_amd.pipeline import GraphedTrainer
n = 4096
q0, hq = scenarios.config2_states(n)
torch.manual_seed(2)
sim = KinovaSim(n, "CubeS", auto_reset=True, horizon=30, contact_tap=True)
obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=torch.device("cuda", 0), capturable=True)
from pathlib import Path
policy.load(str(Path("kinovagrasping_amd/assets/bench_policy/ddpg_256_256")), sync_targets=True)      # the bench's committed policy
replay = DeviceEpisodeReplay(n, capacity=4 * n, horizon=30, device=torch.device("cuda", 0))
eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
eng.start(obs0)
trainer = GraphedTrainer(sim, policy, replay, eng, batch_episodes=64, overlap=False)
trainer.capture()
MODE = sys.argv[1] if len(sys.argv) > 1 else "policy"       # "random": config 2's random actions (bench.py --mode sim) instead of the policy
base = scenarios.config_actions(256, 30)
acts = torch.as_tensor(np.tile(base, (1, 1, n // 256))).cuda()
for t in range(300 if MODE == "policy" else 30):
    trainer.step() if MODE == "policy" else sim.step(acts[t % 30])
tot = np.zeros(4); cnt = 0; per_lane = np.zeros(16); r0 = np.zeros(16); r1 = np.zeros(16)
hist = np.zeros(8)
for t in range(300, 330):
    trainer.step() if MODE == "policy" else sim.step(acts[t % 30])
    st = sim.get_state(contacts=True)
    torch.cuda.synchronize()
    prof = st["contact"].reshape(-1, n)[:480].cpu().numpy().reshape(16, 30, n)       # [lane][slot][env], sums over the env-step's 15 substeps
    lane_pairs, wave_passes, team_lanes = prof[:, 24], prof[:, 25], prof[:, 26]
    live_env = lane_pairs.sum(0)                                                    # live hull pairs of the env, summed over 15 substeps
    tot += [live_env.mean() / 15, wave_passes[0].mean() / 15, lane_pairs.max(0).mean() / 15, np.ceil(live_env / 15 / 16).mean()]
    cnt += 1; per_lane += lane_pairs.mean(1) / 15; r0 += prof[:, 27].mean(1) / 15; r1 += prof[:, 28].mean(1) / 15
print(f"{MODE} regime (policy = the bench's committed policy, lock-step trainer; random = config 2's actions), 30 env-steps x 15 substeps x 4096 envs:")
print(f"  live hull pairs per env and substep (passed both culls): {tot[0] / cnt:.2f}")
print(f"  narrow-phase passes per wave and substep (busiest lane of the 4 envs of the wave): {tot[1] / cnt:.3f}")
print(f"  busiest lane of an env alone: {tot[2] / cnt:.3f}   (a dealing that spreads an env's live pairs over its 16 lanes: 1 pass while an env has <= 16 live pairs)")
print("  live pairs per lane and substep (lanes 0-9 own one hull pair, lanes 10-15 two):", np.round(per_lane / cnt, 3))
GN = ["ground", "palm", "f1_prox", "f1_dist", "f2_prox", "f2_dist", "f3_prox", "f3_dist", "object"]
from kinovagrasping_amd import model_compiler as mc
P = mc.read_blob(scenarios.model_blob("CubeS"))["pairs"]
HARDLY = {(2, 4), (2, 6), (4, 6), (3, 4), (3, 6), (5, 6)}
SELDOM = {(1, 3), (1, 5), (1, 7), (3, 7), (2, 7), (4, 7)}
rar = lambda a, b: 2 if (a, b) in HARDLY else 1 if (a, b) in SELDOM else 0          # ks_model.h: hull_pair_rarity
pairs = [(int(r[0]), int(r[1])) for r in P if int(r[0]) != 0]
if len(sys.argv) > 2 and sys.argv[2] == "index-order":                              # a library built before the re-ordering
    hull = [(GN[a], GN[b]) for a, b in pairs]
else:
    hull = [(GN[a], GN[b]) for c in (0, 1, 2) for a, b in pairs if rar(a, b) == c]    # ks_model.h: model_pair_order
print("  live fraction per hull pair (position in the list the lanes share out: lanes 0 - 15 take 0 - 15, lanes 10 - 15 also 16 - 21):")
for k, (a, b) in enumerate(hull):
    f = r0[k] / cnt if k < 16 else r1[k - 16 + 10] / cnt
    print(f"    {k:2d} {a}-{b}: {f:.3f}")
