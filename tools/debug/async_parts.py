#!/usr/bin/env python3
"""Free-running rollout: how fast is the rollout launch alone, the learner alone (one-wave kernels, KS_MLP_SPLIT=0), and both?
usage (GPU box): python tools/debug/async_parts.py [chunk]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kinovagrasping_amd import scenarios  # noqa: E402
from kinovagrasping_amd.ddpgfd import DDPGfD  # noqa: E402
from kinovagrasping_amd.pipeline import AsyncTrainer  # noqa: E402
from kinovagrasping_amd.replay import DeviceEpisodeReplay  # noqa: E402
from kinovagrasping_amd.rollout import RolloutEngine  # noqa: E402
from kinovagrasping_amd.sim import KinovaSim  # noqa: E402

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = 4096
q0, hq = scenarios.config2_states(n)
sim = KinovaSim(n, "CubeS", horizon=30, auto_reset=True)
obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
torch.manual_seed(2)
policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=sim.device)
replay = DeviceEpisodeReplay(n, capacity=4 * n, horizon=30, device=sim.device)
eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
eng.start(obs0)
tr = AsyncTrainer(sim, policy, replay, eng, batch_episodes=64)
tr.capture()
tr.run(36, learn=False); tr.flush()
for _ in range(60):                      # train a while: contact-rich regime
    tr.run(chunk)
tr.flush(); torch.cuda.synchronize()


def timed(fn, reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    tr.flush(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * chunk) * 1e3


both = timed(lambda: tr.run(chunk), 30)
roll = timed(lambda: tr.run(chunk, learn=False), 30)


def learner_only():
    with torch.cuda.stream(tr.side):
        for _ in range(chunk):
            tr.g_commit.replay(); tr.g_head.replay(); tr.publish(); tr._body()


learn = timed(learner_only, 30)
ph = tr.counters[4:].tolist()
if sum(ph):
    tot = tr.env_steps * (n // 16)
    print("k_rollout phases, mean per workgroup and env-step [us]: policy %.1f  15 substeps %.1f  rays %.1f  observation + replay write %.1f" % tuple(p / tot / 100.0 for p in ph))
print(f"chunk {chunk}: per env-step  rollout + learner {both:.4f} ms   rollout alone {roll:.4f} ms   learner alone {learn:.4f} ms", flush=True)
sim.close()
