"""The free-running rollout kernel (ks_rollout / pipeline.AsyncTrainer): every stepping workgroup loops over its own 16 envs -
in-kernel actor forward + noise + selection rule, 15 substeps, rays, observation, replay write - without waiting for any other
workgroup.  Scheduling is the only thing that changes: per env, the trajectory, the noise stream and the stored transitions are
those of the lock-step calls (kr_actor_select -> ks_step -> kr_store_transition), bit for bit, for the same weights."""
import numpy as np
import pytest
import torch

from kinovagrasping_amd import scenarios

pytestmark = pytest.mark.gpu


def _setup(n, horizon, seed=2, hidden=(256, 256)):
    from kinovagrasping_amd.ddpgfd import DDPGfD
    from kinovagrasping_amd.replay import DeviceEpisodeReplay
    from kinovagrasping_amd.rollout import RolloutEngine
    from kinovagrasping_amd.sim import KinovaSim
    q0, hq = scenarios.config2_states(n)
    sim = KinovaSim(n, "CubeS", horizon=horizon, auto_reset=True)
    obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
    torch.manual_seed(seed)
    policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=hidden, device=sim.device)
    with torch.no_grad():                       # an untrained actor sits at 0.4: push it around so that lifts and early dones happen
        policy.actor.l3.bias.add_(torch.tensor([0.0, 1.0, 0.8, 1.2], device=sim.device))
    replay = DeviceEpisodeReplay(n, capacity=8 * n, horizon=horizon, device=sim.device)
    eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
    eng.start(obs0)
    return sim, policy, replay, eng


def _ring_episodes(replay):
    eps = replay.host_episodes()
    key = lambda e: (len(e["reward"]), e["state"].tobytes(), e["action"].tobytes(), e["next_state"].tobytes(), e["reward"].tobytes(), e["not_done"].tobytes())
    return sorted(key(e) for e in eps)


@pytest.mark.parametrize("hidden", [(256, 256), (64, 64)])
def test_free_running_rollout_equals_the_lock_step_calls(hidden):
    from kinovagrasping_amd.pipeline import AsyncTrainer
    n, horizon, chunks, per = 272, 12, 5, 9              # 17 workgroups; 45 env-steps: every env finishes >= 3 episodes
    # lock step: the three calls per env-step
    sim, policy, replay, eng = _setup(n, horizon, hidden=hidden)
    for _ in range(chunks * per):
        eng.step()
    torch.cuda.synchronize()
    ref = dict(obs=eng.obs.clone(), prev=eng.prev_obs.clone(), t=eng.t.clone(), ready=eng.ready.clone(), qpos=sim.get_state()["qpos"].clone(),
               status=sim.get_state()["status"].clone(), eps=_ring_episodes(replay), count=replay.count, done=eng.done_out.clone())
    sim.close()
    # free running: 5 launches of 9 env-steps, episodes handed over between the launches
    sim, policy, replay, eng = _setup(n, horizon, hidden=hidden)
    tr = AsyncTrainer(sim, policy, replay, eng, batch_episodes=16)
    for _ in range(chunks):
        sim.rollout(per, tr.args)
        replay.commit_published()
    torch.cuda.synchronize()
    st = sim.get_state()
    c = tr.counts()
    print(f"free-running {hidden}: {c}, ring {replay.count} episodes; lock step ring {ref['count']}")
    assert c["episodes_dropped"] == 0 and c["episodes_finished"] >= 3 * n
    assert torch.equal(st["qpos"], ref["qpos"]) and torch.equal(st["status"], ref["status"])
    assert torch.equal(eng.obs, ref["obs"]) and torch.equal(eng.prev_obs, ref["prev"]) and torch.equal(eng.t, ref["t"]) and torch.equal(eng.ready, ref["ready"])
    assert torch.equal(tr.steps_total, torch.full_like(tr.steps_total, chunks * per))
    assert replay.count == ref["count"] == c["episodes_kept"]
    assert _ring_episodes(replay) == ref["eps"]                      # the same episodes, whatever order they arrived in
    sim.close()


def test_async_trainer_trains_beside_the_free_running_rollout():
    from kinovagrasping_amd.pipeline import AsyncTrainer
    n = 512
    sim, policy, replay, eng = _setup(n, 30)
    tr = AsyncTrainer(sim, policy, replay, eng, batch_episodes=64)
    tr.capture()
    w0 = {k: v.clone() for k, v in policy._flat_params.items()}
    tr.run(36, learn=False)                      # every env finishes an episode ...
    tr.flush()                                   # ... and the learner's stream moves them into the ring once the launch is over
    torch.cuda.synchronize()
    assert all(torch.equal(policy._flat_params[k], w0[k]) for k in w0) and replay.count >= n // 2
    for _ in range(3):
        tr.run(30)
    tr.flush(finish_update=True)
    torch.cuda.synchronize()
    c = tr.counts()
    print("async trainer:", c, "updates", tr.updates, "published versions", tr.n_pub)
    assert tr.updates == 90 and tr.n_pub >= 90 and c["episodes_dropped"] == 0 and c["episodes_finished"] >= 3 * n
    assert int(tr.pub_ver) == tr.n_pub and torch.equal(tr.pub[tr.n_pub % 3, :tr.actor_flat.numel()], tr.actor_flat)
    for k in ("actor", "critic", "critic_target"):
        w = policy._flat_params[k]
        assert torch.isfinite(w).all() and (w - w0[k]).abs().max().item() > 0, k
    assert torch.isfinite(tr.native.losses).all()
    assert (sim.get_state()["status"] & 2).sum().item() == 0 and torch.equal(tr.steps_total, torch.full_like(tr.steps_total, 126))
    sim.close()
