#!/usr/bin/env python3
"""GPU box: does work on the learner's stream run BESIDE the persistent rollout kernel?  usage: overlap_probe.py N [cubes|mixed] [nshapes]"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kinovagrasping_amd import scenarios
from kinovagrasping_amd.ddpgfd import DDPGfD
from kinovagrasping_amd.pipeline import AsyncTrainer
from kinovagrasping_amd.replay import DeviceEpisodeReplay
from kinovagrasping_amd.rollout import RolloutEngine
from kinovagrasping_amd.sim import KinovaSim
n = int(sys.argv[1]); mixed = sys.argv[2] == "mixed"; nshapes = int(sys.argv[3]) if len(sys.argv) > 3 else 14
if mixed:
    oid, _, q0, hq, mf = scenarios.config5_states(n, seed=5, cohort=16)
    oid = (oid % nshapes).astype(np.int32)
    sim = KinovaSim(n, scenarios.SHAPES[:nshapes], horizon=30, auto_reset=True)
    obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq), object_id=oid, mass_friction=mf)
else:
    q0, hq = scenarios.config2_states(n)
    sim = KinovaSim(n, "CubeS", horizon=30, auto_reset=True)
    obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
torch.manual_seed(2)
policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=sim.device, capturable=True)
replay = DeviceEpisodeReplay(n, capacity=4 * n, horizon=30, device=sim.device)
eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
eng.start(obs0)
tr = AsyncTrainer(sim, policy, replay, eng, batch_episodes=64)
tr.capture()
tr.run(36, learn=False); tr.flush(); torch.cuda.synchronize()
x = torch.zeros(1 << 16, device=sim.device)
for what in ("elementwise add", "g_commit", "g_head", "g_learn[0]", "publish"):
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    tr.side.wait_stream(tr.main); tr.main.wait_stream(tr.side)
    e0.record(tr.main)
    sim.rollout(40, tr.args)
    e2.record(tr.main)
    with torch.cuda.stream(tr.side):
        for _ in range(5):
            if what == "elementwise add": x.add_(1.0)
            elif what == "g_commit": tr.g_commit.replay()
            elif what == "g_head": tr.g_head.replay()
            elif what == "publish": tr.publish()
            else: tr.g_learn[0].replay()
        e1.record(tr.side)
    torch.cuda.synchronize()
    print(f"{what:16s}: 5 x on the learner's stream done {e0.elapsed_time(e1):8.2f} ms after the rollout launch was issued; the launch (40 env-steps) took {e0.elapsed_time(e2):8.2f} ms")
    replay.commit_published(); torch.cuda.synchronize()
