#!/usr/bin/env python3
"""Dev-only (GPU box): rollout rate of ONE context holding the 14 shape keys of the reference's stage-2 experiments (main_DDPGfD.py:1270-1281: single- and
multi-geom objects, libkinova_sim_mg.so), 4096 envs, actor in the loop + replay writes: lock step (kr_actor_select -> ks_step -> kr_store_transition) against the
free-running rollout kernel (ks_rollout, round-robin dealing of the 16-env groups), no learner."""
import sys, time, warnings
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kinovagrasping_amd import curriculum, scenarios
from kinovagrasping_amd.ddpgfd import DDPGfD
from kinovagrasping_amd.pipeline import AsyncTrainer
from kinovagrasping_amd.replay import DeviceEpisodeReplay
from kinovagrasping_amd.rollout import RolloutEngine
from kinovagrasping_amd.sim import KinovaSim

n, horizon = 4096, 30
shapes = curriculum.experiment_plan(4)["requested_shapes"]
assert len(shapes) == 14


def setup():
    sim = KinovaSim(n, shapes, horizon=horizon, auto_reset=True)
    per = n // 16 // len(shapes) * 16                                    # whole 16-env groups per shape, the rest to the first shapes
    oid = np.repeat(np.arange(len(shapes)), per)
    oid = np.concatenate([oid, np.repeat(np.arange(len(shapes)), 16)[: n - len(oid)]]).astype(np.int32)
    oid.sort()
    rng = np.random.RandomState(3)
    q = np.zeros((16, n)); q[12] = 1
    hq = np.zeros((4, n))
    for e in range(n):
        sh = shapes[oid[e]]
        o = scenarios.select_orientation(sh, "random", rng)
        cmd = scenarios.start_coord_table(sh, o)[rng.randint(0, 4000)] if scenarios.has_start_table(sh, o) else scenarios.fallback_start(sh, o, rng)
        q[9:12, e] = scenarios.reset_body_position(sh, cmd)
        q[0:3, e] = scenarios.hand_slide_offsets(o, sh, "pose")
        hq[:, e] = scenarios.hand_quat_for(o)
    obs0 = sim.reset(torch.as_tensor(q), torch.as_tensor(hq), object_id=oid)
    torch.manual_seed(3)
    policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=sim.device)
    with torch.no_grad():
        policy.actor.l3.bias.add_(torch.tensor([-6.0, 1.0, 0.8, 1.2], device=sim.device))       # closes the hand, check_grasp fires, scripted lift
    replay = DeviceEpisodeReplay(n, capacity=4 * n, horizon=horizon, device=sim.device)
    eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
    eng.start(obs0)
    return sim, policy, replay, eng


sim, policy, replay, eng = setup()
for _ in range(30):
    eng.step()
torch.cuda.synchronize()
t0 = time.time()
K = 90
for _ in range(K):
    eng.step()
torch.cuda.synchronize()
dt = time.time() - t0
st = sim.get_state()
print(f"lock step   : {n * K / dt / 1e6:.3f} M env-steps/s ({dt / K * 1e3:.3f} ms per env-step), status {sorted(set(st['status'].cpu().numpy().tolist()))}, contacts per env {float(st['ncon'].float().mean()):.2f}")
sim.close()
sim, policy, replay, eng = setup()
with warnings.catch_warnings():
    warnings.simplefilter("ignore", RuntimeWarning)
    tr = AsyncTrainer(sim, policy, replay, eng, batch_episodes=16)
sim.rollout(30, tr.args); replay.commit_published()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(3):
    sim.rollout(30, tr.args)
    replay.commit_published()
torch.cuda.synchronize()
dt = time.time() - t0
st = sim.get_state()
print(f"free running: {n * K / dt / 1e6:.3f} M env-steps/s ({dt / K * 1e3:.3f} ms per env-step), {tr.counts()}, status {sorted(set(st['status'].cpu().numpy().tolist()))}")
# round 6, opt-in: launches with a TIME budget (ks_rollout_args.budget_ticks) - every wave steps its four envs until the budget has passed, so the cubes are
# not paced by the bowls; the rate is the env-steps actually done per second, and the object mix of what is collected is uneven (printed)
print("rollout plan:", sim.rollout_plan())
if sim.rollout_plan()[0] == "waves":
    s0 = tr.steps_total.clone()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(6):
        tr.args.budget_ticks = 6_000_000                     # 60 ms per launch
        sim.rollout(200, tr.args)
        tr.args.budget_ticks = 0
        replay.commit_published()
    torch.cuda.synchronize()
    dt = time.time() - t0
    d = (tr.steps_total - s0).cpu().numpy()
    st = sim.get_state()
    oid = sim.get_state().get("obj_id") if isinstance(sim.get_state(), dict) and "obj_id" in sim.get_state() else None
    per = n // 16 // len(shapes) * 16
    oid = np.repeat(np.arange(len(shapes)), per)
    oid = np.sort(np.concatenate([oid, np.repeat(np.arange(len(shapes)), 16)[: n - len(oid)]]))
    mix = {shapes[k]: round(float(d[oid == k].mean()), 1) for k in range(len(shapes))}
    print(f"time-budgeted (6 launches of 60 ms): {d.sum() / dt / 1e6:.3f} M env-steps/s; env-steps per env min {d.min()} mean {d.mean():.1f} max {d.max()}; mean per object {mix}; "
          f"{tr.counts()}, status {sorted(set(st['status'].cpu().numpy().tolist()))}")
