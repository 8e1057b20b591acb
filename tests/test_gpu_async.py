"""The free-running rollout kernel (ks_rollout / pipeline.AsyncTrainer): every stepping wave loops over its own 4 envs (round 6; with more 16-env
groups than compute units: every workgroup over its groups, from a ready queue or a fixed deal) - in-kernel actor forward + noise + selection
rule, 15 substeps, rays, observation, replay write - without waiting for any other wave or workgroup.  Scheduling is the only thing that changes: per env, the trajectory, the noise stream and the stored transitions are
those of the lock-step calls (kr_actor_select -> ks_step -> kr_store_transition), bit for bit, for the same weights."""
import numpy as np
import pytest
import torch

from kinovagrasping_amd import scenarios

pytestmark = pytest.mark.gpu


def _setup(n, horizon, seed=2, hidden=(256, 256), mixed=False, cohort=1):
    from kinovagrasping_amd.ddpgfd import DDPGfD
    from kinovagrasping_amd.replay import DeviceEpisodeReplay
    from kinovagrasping_amd.rollout import RolloutEngine
    from kinovagrasping_amd.sim import KinovaSim
    if mixed:            # BASELINE config 5's start states: 14 objects x 3 hand poses x mass / friction in one context
        oid, _, q0, hq, mf = scenarios.config5_states(n, seed=5, cohort=cohort)
        sim = KinovaSim(n, scenarios.SHAPES, horizon=horizon, auto_reset=True)
        obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq), object_id=oid, mass_friction=mf)
    else:
        q0, hq = scenarios.config2_states(n)
        sim = KinovaSim(n, "CubeS", horizon=horizon, auto_reset=True)
        obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
    torch.manual_seed(seed)
    policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=hidden, device=sim.device)
    with torch.no_grad():                       # an untrained actor sits at 0.4: push it around so that lifts and early dones happen
        policy.actor.l3.bias.add_(torch.tensor([-6.0, 1.0, 0.8, 1.2], device=sim.device))    # wrist ~ 0, fingers ~ 0.6: closes, check_grasp fires, scripted lift
    replay = DeviceEpisodeReplay(n, capacity=8 * n, horizon=horizon, device=sim.device)
    eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
    eng.start(obs0)
    return sim, policy, replay, eng


def _ring_episodes(replay):
    eps = replay.host_episodes()
    key = lambda e: (len(e["reward"]), e["state"].tobytes(), e["action"].tobytes(), e["next_state"].tobytes(), e["reward"].tobytes(), e["not_done"].tobytes())
    return sorted(key(e) for e in eps)


@pytest.mark.parametrize("hidden,mixed,horizon,per,n", [((256, 256), False, 12, 9, 272), ((64, 64), False, 12, 9, 272), ((256, 256), True, 12, 9, 272),
                                                        ((256, 256), False, 30, 13, 272), ((256, 256), False, 30, 12, 4096),
                                                        ((256, 256), "cohort", 30, 7, 8192), ((256, 256), True, 12, 5, 4400)])
def test_free_running_rollout_equals_the_lock_step_calls(hidden, mixed, horizon, per, n):
    _free_running_equals_lock_step(hidden, mixed, horizon, per, n)


@pytest.mark.parametrize("env,n,mixed,plan", [({"KS_ROLLOUT_WAVES": "0"}, 272, False, "workgroups"), ({"KS_ROLLOUT_DEAL": "static"}, 4400, True, "runs"),
                                              ({"KS_ROLLOUT_DEAL": "rr"}, 4400, True, "round-robin"), ({}, 4400, True, "queue"), ({}, 272, False, "waves"),
                                              ({"KS_ROLLOUT_PHASE_DEAL": "0"}, 272, False, "waves"), ({"KS_ROLLOUT_PHASE_DEAL": "2"}, 272, False, "waves")])
def test_every_scheduling_form_of_the_rollout_kernel_equals_lock_step(monkeypatch, env, n, mixed, plan):
    """the fixed deals of rounds 3-4 (KS_ROLLOUT_DEAL=static / rr - still the multi-geom library's default) and the barrier-joined
    workgroup form (KS_ROLLOUT_WAVES=0) are no longer anybody's default in the standard library: exercised here explicitly, with the library's own
    report of what it runs (ks_rollout_plan).  Round 6: the free waves' slot list in env order / sorted by episode step over the whole list
    (KS_ROLLOUT_PHASE_DEAL=0 / 2; the default - sorted within every workgroup - runs in every "waves" case): a slot only says which lanes step an env."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _free_running_equals_lock_step((256, 256), mixed, 12, 5, n, expect_plan=plan)


def _free_running_equals_lock_step(hidden, mixed, horizon, per, n, expect_plan=None):
    """horizon 12: every env runs into the time limit three times in 45 env-steps; horizon 30, 65 env-steps: the (bias-pushed) actor closes
    the hand, check_grasp fires, the scripted lift ends episodes early - the un-stored lift steps and the overwrite of the last stored
    transition (utils.py:309-343) are part of what must match.  n = 4096: the bench's shape (BASELINE config 3: one workgroup on every CU).
    n = 8192, 14 shapes in 16-env cohorts: BASELINE config 5's per-GPU shape as bench.py runs it - 512 groups, every one of the 256 persistent
    workgroups steps two of them in turn and restages another object's tables in between; n = 4400 mixed: 288 groups that do NOT divide
    evenly (some workgroups step two groups, most one)."""
    from kinovagrasping_amd.pipeline import AsyncTrainer
    import warnings
    chunks = 5                                           # n = 272: 17 workgroups
    cohort, mixed = (16 if mixed == "cohort" else 1), bool(mixed)
    # lock step: the three calls per env-step
    sim, policy, replay, eng = _setup(n, horizon, hidden=hidden, mixed=mixed, cohort=cohort)
    for _ in range(chunks * per):
        eng.step()
    torch.cuda.synchronize()
    ref = dict(obs=eng.obs.clone(), prev=eng.prev_obs.clone(), t=eng.t.clone(), ready=eng.ready.clone(), qpos=sim.get_state()["qpos"].clone(),
               status=sim.get_state()["status"].clone(), eps=_ring_episodes(replay), count=replay.count, done=eng.done_out.clone())
    sim.close()
    # free running: 5 launches of 9 env-steps, episodes handed over between the launches
    sim, policy, replay, eng = _setup(n, horizon, hidden=hidden, mixed=mixed, cohort=cohort)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)          # (the uneven 4400-env case warns about its imbalance)
        tr = AsyncTrainer(sim, policy, replay, eng, batch_episodes=16)
    if expect_plan is not None:
        assert sim.rollout_plan()[0] == expect_plan == tr.rollout_plan, (sim.rollout_plan(), expect_plan)
    for _ in range(chunks):
        sim.rollout(per, tr.args)
        replay.commit_published()
    torch.cuda.synchronize()
    st = sim.get_state()
    c = tr.counts()
    print(f"free-running {hidden} mixed={mixed}: {c}, ring {replay.count} episodes; lock step ring {ref['count']}")
    assert c["episodes_dropped"] == 0 and c["episodes_finished"] >= ((3 if per >= 9 else 2) if horizon == 12 else (2 if per >= 12 else 1)) * n
    if horizon == 30 and not mixed:
        assert c["lifted"] > 0.05 * n
    assert torch.equal(st["qpos"], ref["qpos"]) and torch.equal(st["status"], ref["status"])
    assert torch.equal(eng.obs, ref["obs"]) and torch.equal(eng.prev_obs, ref["prev"]) and torch.equal(eng.t, ref["t"]) and torch.equal(eng.ready, ref["ready"])
    assert torch.equal(tr.steps_total, torch.full_like(tr.steps_total, chunks * per))
    assert replay.count == ref["count"] == min(c["episodes_kept"], replay.capacity)
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


def test_episodes_committed_beside_the_running_kernel_are_whole():
    """BASELINE config 3's shape (4096 envs, one workgroup on every CU, launches of 60 env-steps): the learner's stream moves published episodes into the
    ring WHILE the rollout kernel keeps writing the envs' other open buffer (release store of pub_len in k_rollout, plain loads in the commit kernels of
    later launches, possibly on another XCD).  Every committed episode must be one env's consecutive transitions: next_state[t] == state[t + 1] bit for
    bit, not_done 1 except on the last row, rewards 0 except possibly the last - a row of another episode (stale line, torn hand-over) breaks that."""
    from kinovagrasping_amd.pipeline import AsyncTrainer
    n = 4096
    sim, policy, replay, eng = _setup(n, 30)
    tr = AsyncTrainer(sim, policy, replay, eng, batch_episodes=64)
    tr.capture()
    tr.run(36, learn=False)
    tr.flush()
    for _ in range(3):
        tr.run(60)
    tr.flush(finish_update=True)
    torch.cuda.synchronize()
    c = tr.counts()
    assert c["episodes_dropped"] == 0 and c["episodes_kept"] >= 6 * n and replay.count == min(c["episodes_kept"], replay.capacity), (c, replay.count)
    L = replay.ep_len[: replay.count]
    S, NX, R, ND = replay.ep_state[: replay.count], replay.ep_next[: replay.count], replay.ep_reward[: replay.count], replay.ep_not_done[: replay.count]
    H = S.shape[1]
    t = torch.arange(H, device=S.device)[None, :]
    inner = t < (L[:, None] - 1)                               # rows with a successor in the same episode
    assert L.min().item() > replay.n_steps + 1 and L.max().item() <= H
    chain = (NX[:, :-1] == S[:, 1:]).all(2) | ~inner[:, :-1]
    bad = (~chain).any(1)
    assert not bad.any(), f"{int(bad.sum())} of {replay.count} committed episodes are not one env's consecutive transitions"
    assert ((ND == 1) | ~inner).all() and (ND.gather(1, (L - 1)[:, None]) == 0).all()
    assert ((R == 0) | ~inner).all()
    print(f"{replay.count} committed episodes whole; lengths {L.min().item()}..{L.max().item()}, lifted {c['lifted']}")
    sim.close()


def test_long_launches_drop_no_episode():
    """One launch of 240 env-steps with the learner beside it: the learner's stream (the faster one) is paced on the envs' step
    counters (kr_wait_min), so the episodes published late in the launch are still collected - without pacing the learner is done
    ~45 env-steps before the rollout and envs run out of open buffers (61 episodes dropped in the bench's 240-step run)."""
    from kinovagrasping_amd.pipeline import AsyncTrainer
    n = 4096
    sim, policy, replay, eng = _setup(n, 30)
    tr = AsyncTrainer(sim, policy, replay, eng, batch_episodes=64)
    tr.capture()
    tr.run(36, learn=False)
    tr.flush()
    tr.run(240)
    tr.flush(finish_update=True)
    torch.cuda.synchronize()
    c = tr.counts()
    print("240-step launch:", c)
    assert torch.equal(tr.steps_total, torch.full_like(tr.steps_total, 276)) and tr.updates == 240
    assert c["episodes_dropped"] == 0 and c["episodes_finished"] >= 9 * n
    sim.close()


def test_wait_min_kernel_holds_its_stream_until_every_counter_has_arrived_or_the_clock_runs_out():
    """kr_wait_min (include/kinova_rollout.h): what follows it on a stream starts when min(values) >= target - or after the time-out."""
    import time
    from kinovagrasping_amd.sim import load_library
    L = load_library()
    dev = torch.device("cuda", 0)
    vals = torch.zeros(300, dtype=torch.long, device=dev)
    flag = torch.zeros(1, device=dev)
    waiter, writer = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(waiter):
        assert L.kr_wait_min(vals.data_ptr(), 300, 5, 20.0, waiter.cuda_stream) == 0
        flag.add_(1.0)
        done = torch.cuda.Event()
        done.record(waiter)
    time.sleep(0.05)
    assert not done.query()                                   # still waiting: nobody has counted yet
    with torch.cuda.stream(writer):
        vals[:299] += 7                                       # all but one
    writer.synchronize()
    time.sleep(0.05)
    assert not done.query()                                   # the minimum is what counts
    with torch.cuda.stream(writer):
        vals[299:] += 5
    writer.synchronize()
    t0 = time.perf_counter()
    while not done.query() and time.perf_counter() - t0 < 5.0:
        time.sleep(0.001)
    assert done.query() and flag.item() == 1.0
    # the time-out: a target that never arrives releases the stream after ~0.2 s
    t0 = time.perf_counter()
    with torch.cuda.stream(waiter):
        assert L.kr_wait_min(vals.data_ptr(), 300, 1000, 0.2, waiter.cuda_stream) == 0
    waiter.synchronize()
    assert 0.15 < time.perf_counter() - t0 < 2.0
    assert L.kr_wait_min(None, 300, 1, 1.0, None) != 0 and L.kr_wait_min(vals.data_ptr(), 0, 1, 1.0, None) != 0


def test_trainer_picks_a_learner_stream_that_overlaps_the_rollout_stream():
    """torch hands out pooled streams round-robin and the GPU has few hardware queues: after other code has created streams, a new one can
    share the rollout stream's queue and the learner would run BEHIND the persistent kernel (episodes dropped).  The trainers probe."""
    from kinovagrasping_amd.pipeline import AsyncTrainer
    junk = [torch.cuda.Stream(torch.device("cuda", 0)) for _ in range(37)]          # shift the pool's round-robin
    sim, policy, replay, eng = _setup(256, 30)
    tr = AsyncTrainer(sim, policy, replay, eng, batch_episodes=16)
    assert tr.streams_overlap is True
    del junk
    sim.close()


def test_time_budgeted_launch_advances_envs_by_time_on_their_lock_step_trajectories():
    """ks_rollout_args.budget_ticks (opt-in, round 6): a wave starts no further env-step once the launch's time budget has passed, so envs do between 1
    and n_iter env-steps - and every env is exactly where the lock-step calls put it after as many env-steps as it did (same arithmetic, the noise
    keyed by the env's own step count).  The workgroup forms of the kernel refuse a budget."""
    from kinovagrasping_amd.pipeline import AsyncTrainer
    n, horizon, K = 1024, 30, 14
    sim, policy, replay, eng = _setup(n, horizon)
    hist = [sim.get_state()["qpos"].clone()]
    obs_hist = [eng.obs.clone()]
    for _ in range(K):
        eng.step()
        hist.append(sim.get_state()["qpos"].clone()); obs_hist.append(eng.obs.clone())
    torch.cuda.synchronize()
    sim.close()
    sim, policy, replay, eng = _setup(n, horizon)
    tr = AsyncTrainer(sim, policy, replay, eng, batch_episodes=16)
    assert sim.rollout_plan()[0] == "waves"
    tr.args.budget_ticks = 600_000                          # 6 ms of the 100 MHz clock: about half of what 14 env-steps take
    sim.rollout(K, tr.args)
    tr.args.budget_ticks = 0
    torch.cuda.synchronize()
    done = tr.steps_total.cpu().numpy()
    print(f"time-budgeted launch: env-steps per env min {done.min()} mean {done.mean():.2f} max {done.max()} of at most {K}")
    assert done.min() >= 1 and done.max() <= K and done.min() < done.max() and (done.reshape(-1, 4) == done.reshape(-1, 4)[:, :1]).all()   # (the four envs of a wave advance together)
    q = sim.get_state()["qpos"]
    H, OH = torch.stack(hist), torch.stack(obs_hist)        # [K + 1, 16, n], [K + 1, n, 82]
    idx = torch.as_tensor(done, device=q.device)
    assert torch.equal(q, H[idx, :, torch.arange(n, device=q.device)].T)
    assert torch.equal(eng.obs, OH[idx, torch.arange(n, device=q.device)])
    assert (sim.get_state()["status"] == 0).all() and tr.counts()["episodes_dropped"] == 0
    sim.close()
    # a context with more groups than compute units (ready queue): refused
    sim, policy, replay, eng = _setup(4400, 12, mixed=True)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        tr = AsyncTrainer(sim, policy, replay, eng, batch_episodes=16)
    tr.args.budget_ticks = 1000
    with pytest.raises(RuntimeError):
        sim.rollout(3, tr.args)
    sim.close()
