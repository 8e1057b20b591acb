#!/usr/bin/env python3
"""Dev tool (GPU box, -DKS_STAMP -DKS_STAMP_SPLIT build via KS_LIB): the hull-pair loop of the two-lane build in the bench's policy regime - wave-level cycles of
the distance-query part and of the penetration part of a pass, queries and turns per env-step."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
exec(open('tools/r06/env_chains.py').read().split("n = 4096")[0])
n = 4096
q0, hq = scenarios.config2_states(n)
torch.manual_seed(2)
sim = KinovaSim(n, "CubeS", auto_reset=True, horizon=30, contact_tap=True)
obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=torch.device("cuda", 0), capturable=True)
policy.load(str(Path("kinovagrasping_amd/assets/bench_policy/ddpg_256_256")), sync_targets=True)
replay = DeviceEpisodeReplay(n, capacity=4 * n, horizon=30, device=torch.device("cuda", 0))
eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
eng.start(obs0)
trainer = GraphedTrainer(sim, policy, replay, eng, batch_episodes=64, overlap=False)
trainer.capture()
for t in range(300):
    trainer.step()
acc = []
for t in range(40):
    trainer.step()
    st = sim.get_state(contacts=True)
    torch.cuda.synchronize()
    prof = st["contact"].reshape(-1, n)[:480].cpu().numpy().reshape(16, 30, n)
    # lane 0 of every env carries the wave-level stamps (every lane of a wave sees the same elapsed cycles)
    acc.append([prof[0, 6].mean(), prof[0, 9].mean(), prof[0, 27].mean(), prof[0, 28].mean(), prof[:, 24].sum(0).mean(), prof[:, 25].sum(0).mean(), prof[0, 26].mean(),
                prof[:, 29].sum(0).mean(), prof[:, 29].max(0).mean()])
a = np.mean(acc, 0)
print(f"per env-step (k cycles of a wave): total {a[0]/1e3:.0f}, hull-pair phase {a[1]/1e3:.0f} = distance-query part {a[2]/1e3:.0f} + penetration part {a[3]/1e3:.0f} + the rest (culls, dealing) {(a[1]-a[2]-a[3])/1e3:.0f}")
print(f"per env and env-step: live pairs {a[4]:.1f}, penetration queries {a[5]:.2f}, passes of the wave with a penetration query {a[6]:.1f} of 15, turns of the env's queries {a[7]:.1f} (busiest lane {a[8]:.1f})")
print(f"-> {a[3]/max(a[6],1e-9):.0f} cycles of penetration part per pass that has one")

