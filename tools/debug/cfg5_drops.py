#!/usr/bin/env python3
"""GPU box: why does the free-running rollout drop episodes at 8192 envs (two groups per persistent workgroup)?"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
mixed = (sys.argv[2] if len(sys.argv) > 2 else "mixed") == "mixed"
cohort = int(sys.argv[3]) if len(sys.argv) > 3 else 16
nshapes = int(sys.argv[4]) if len(sys.argv) > 4 else 14
if mixed:
    oid, _, q0, hq, mf = scenarios.config5_states(n, seed=5, cohort=cohort)
    if nshapes < 14:
        # fewer objects: shapes folded onto the first `nshapes` (start rows / poses stay those drawn for the original shape)
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
if len(tr.counters) > 8 + 4 * 512 + 8:
    tr.counters[6] = 1 << 40
tr.capture()
tr.run(36, learn=False); tr.flush(); torch.cuda.synchronize()
prev = tr.counts()
for it in range(6):
    t0 = time.perf_counter()
    e_main, e_side = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0 = torch.cuda.Event(enable_timing=True); e0.record(tr.main)
    tr.run(60)
    e_main.record(tr.main); e_side.record(tr.side)
    if it == 3:
        # watch the ring fill while the launch runs: ring count and the envs' progress, read on a third stream every ~5 ms
        third = torch.cuda.Stream(sim.device)
        host = torch.zeros(4, dtype=torch.long).pin_memory()
        trace = []
        t_poll = time.perf_counter()
        while not e_main.query():
            with torch.cuda.stream(third):
                host[0:1].copy_(replay._count.reshape(1), non_blocking=True)
                host[1:2].copy_(tr.steps_total.min().reshape(1), non_blocking=True)
                host[2:3].copy_(tr.steps_total.max().reshape(1), non_blocking=True)
                host[3:4].copy_((replay.pub_len > 0).sum().reshape(1), non_blocking=True)
            third.synchronize()
            trace.append((round((time.perf_counter() - t_poll) * 1e3, 1), *host.tolist()))
            time.sleep(0.004)
        print("   [ms since poll start, ring count, min steps, max steps, published buffers waiting]:", trace[::2])
    tr.flush(); torch.cuda.synchronize()
    print(f"   rollout stream done after {e0.elapsed_time(e_main):.1f} ms, learner stream after {e0.elapsed_time(e_side):.1f} ms")
    dt = time.perf_counter() - t0
    c = tr.counts()
    st = tr.steps_total
    print(f"launch {it}: {dt / 60 * 1e3:.3f} ms/env-step; finished +{c['episodes_finished'] - prev['episodes_finished']} kept +{c['episodes_kept'] - prev['episodes_kept']} "
          f"dropped +{c['episodes_dropped'] - prev['episodes_dropped']} timeouts {c['pacing_timeouts']}; steps_total min {int(st.min())} max {int(st.max())}; "
          f"pub_len>0: {int((replay.pub_len > 0).sum())}, both buffers published: {int(((replay.pub_len[0] > 0) & (replay.pub_len[1] > 0)).sum())}")
    prev = c
    if len(tr.counters) > 8 + 4 * 512 + 8:
        d = tr.counters[4:8].tolist()
        print(f"   drops {d[1]}: mean env-steps since the env's previous publication {d[0] / max(1, d[1]):.1f}, smallest {d[2]}; other buffer free on a second look: {d[3]}")
