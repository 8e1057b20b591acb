"""Dev tool: per-step duration of k_env_step (HIP events, ks_kernel_time) over the bench's DDPG workload, serial learner.
usage: python tools/step_times.py [steps]   (KS_LIB selects the library build)"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from kinovagrasping_amd import scenarios
from kinovagrasping_amd.ddpgfd import DDPGfD
from kinovagrasping_amd.pipeline import GraphedTrainer
from kinovagrasping_amd.replay import DeviceEpisodeReplay
from kinovagrasping_amd.rollout import RolloutEngine
from kinovagrasping_amd.sim import KinovaSim

n = 4096
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device("cuda", 0)
q0, hq = scenarios.config2_states(n)
sim = KinovaSim(n, "CubeS", auto_reset=True, horizon=30)
obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
torch.manual_seed(2)
policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=dev, capturable=True)
replay = DeviceEpisodeReplay(n, capacity=4 * n, horizon=30, device=dev)
eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
eng.start(obs0)
tr = GraphedTrainer(sim, policy, replay, eng, batch_episodes=64, overlap=False)
tr.capture()
ms = []
for t in range(steps):
    sim.kernel_time(reset=True)
    tr.step()
    ms.append(sim.kernel_time()[0])
ms = np.array(ms)
print("per-step k_env_step ms, 10 per row (step 0 = first after the 3 capture warm-up steps):")
for a in range(0, steps, 10):
    print(f"{a:4d}: " + " ".join(f"{x:6.3f}" for x in ms[a:a + 10]))
print(f"mean over the last 60 steps: {ms[-60:].mean():.4f} ms")
