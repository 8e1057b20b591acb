#!/usr/bin/env python3
"""Dev tool (GPU box, -DKS_STAMP build via KS_LIB): narrow-phase PASSES of a wave per substep in the bench's policy regime - how often a lane of the wave
holds two live hull pairs at once (the second pass of the hull-pair loop), and the phase split of an env-step."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
exec(open('tools/r06/env_chains.py').read().split("newton, hull, passes")[0].split('steps = int')[0])
steps = 40
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
P, L, L0, L1, ph = [], [], [], [], []
for t in range(steps):
    trainer.step()
    st = sim.get_state(contacts=True)
    torch.cuda.synchronize()
    prof = st["contact"].reshape(-1, n)[:480].cpu().numpy().reshape(16, 30, n)
    P.append(prof[0, 25].copy()); L.append(prof[:, 24].sum(0)); L0.append(prof[:, 27].sum(0)); L1.append(prof[:, 28].sum(0)); ph.append(prof[0, :24].mean(1))
P, L, L0, L1 = (np.array(x) for x in (P, L, L0, L1))
print(f"per env-step of 15 substeps: narrow-phase passes of the env's wave {P.mean():.2f} (a pass in every substep would be 15); live hull pairs of an env {L.mean():.2f}, of them dealt in round 0 {L0.mean():.2f}, round 1 {L1.mean():.2f}")
print("phase stamps (k cycles per env-step, wave), index: value:", {i: round(float(v) / 1e3) for i, v in enumerate(np.mean(ph, 0)) if v > 0})
