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
ph = tr.counters[4:8].tolist()
if sum(ph):
    import numpy as np
    nwg = n // 16
    for label, fn in (("rollout alone", lambda: tr.run(chunk, learn=False)), ("rollout + learner", lambda: tr.run(chunk))):
        tr.flush(); torch.cuda.synchronize()
        ph0, st0, rp0 = tr.counters[4:8].clone(), tr.env_steps, tr.counters[8 + 4 * 512:].clone()
        for _ in range(3):
            fn()
        tr.flush(); torch.cuda.synchronize()
        dph = ((tr.counters[4:8] - ph0).double() / ((tr.env_steps - st0) * nwg) / 100.0).tolist()
        print(f"{label}: phases, mean per workgroup and env-step [us]: policy %.1f  15 substeps %.1f  rays %.1f  observation + replay write %.1f" % tuple(dph))
        rp = ((tr.counters[8 + 4 * 512:] - rp0).double() / ((tr.env_steps - st0) * nwg)).tolist()
        print(f"{label}: rays per workgroup and env-step: snapshot + culling %.1f us, walks %.1f us; surviving (ray, geom) tasks %.0f of 2176, node visits %.0f "
              f"(= %.1f per lane), the busiest lane's visits %.0f, subtrees handed to waiting lanes %.0f" % (rp[0] / 100.0, rp[1] / 100.0, rp[2], rp[3], rp[3] / 256.0, rp[4], rp[5]))
        raw = tr.counters[8:8 + 4 * 512].cpu().numpy().reshape(4, 512)[:, :nwg].astype(np.float64)
        mhz = raw[3] / (raw[2] / 100.0)
        print(f"{label}: shader clock during the loop (cycles / wall time) mean {mhz.mean():.0f} MHz  min {mhz.min():.0f}  max {mhz.max():.0f}")
        c = raw[:3] / 100.0          # us
        entry, stage, loop = c[0] - c[0].min(), c[1], c[2]
        end = entry + stage + loop
        print(f"{label}, last launch of {chunk} env-steps, per workgroup [us]: entry spread max {entry.max():.0f}; tables -> LDS mean {stage.mean():.0f} max {stage.max():.0f}; "
              f"loop mean {loop.mean():.0f}  p50 {np.median(loop):.0f}  p90 {np.percentile(loop, 90):.0f}  max {loop.max():.0f}; first entry -> last exit {end.max():.0f} "
              f"= {end.max() / chunk:.0f} per env-step (mean loop / step {loop.mean() / chunk:.0f})")
if sum(ph):
    tot = tr.env_steps * (n // 16)
    print("k_rollout phases, mean per workgroup and env-step [us]: policy %.1f  15 substeps %.1f  rays %.1f  observation + replay write %.1f" % tuple(p / tot / 100.0 for p in ph))
print(f"chunk {chunk}: per env-step  rollout + learner {both:.4f} ms   rollout alone {roll:.4f} ms   learner alone {learn:.4f} ms", flush=True)
sim.close()
