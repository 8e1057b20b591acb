"""Learner state across checkpoints and graph capture (GPU): the native DDPGfD update keeps its Adam moments in flat
buffers of its own - they must reach the reference's 4-file checkpoint (DDPGfD.py:371-382 saves both optimizers) and
come back from it; and GraphedTrainer.capture()'s eager warm-up must not train."""
import numpy as np
import pytest
import torch

from kinovagrasping_amd import scenarios

pytestmark = pytest.mark.gpu


def _batches(dev, k, seed=3, R=320, n=5):
    g = torch.Generator(device=dev).manual_seed(seed)
    out = []
    for _ in range(k):
        out.append((torch.randn(R, n, 82, device=dev, generator=g) * 0.3, torch.rand(R, n, 4, device=dev, generator=g) * 0.8,
                    torch.randn(R, n, 82, device=dev, generator=g) * 0.3, torch.rand(R, n, device=dev, generator=g) * 5,
                    (torch.rand(R, device=dev, generator=g) < 0.8).float()))
    return out


@pytest.mark.parametrize("hidden", [(256, 256), (400, 300)])
def test_checkpoint_roundtrip_keeps_the_adam_state(tmp_path, hidden):
    """6 updates -> save -> load into a fresh policy -> 6 more (crossing the soft target update of the 10th): the native
    learner, the autograd learner, and a native checkpoint continued by the autograd learner all end at the same weights
    (same bound as the native-vs-autograd test); the optimizer files hold moments and step 6."""
    from kinovagrasping_amd.ddpgfd import DDPGfD
    from kinovagrasping_amd.learner_native import NativeDDPGfDUpdate
    dev = torch.device("cuda", 0)
    bs = _batches(dev, 12)

    def fresh():
        torch.manual_seed(7)
        return DDPGfD(82, 4, 0.8, 5, hidden=hidden, device=dev)

    # reference run: autograd, no checkpoint in between
    ref = fresh()
    for b in bs:
        ref.train_on_batch(*b)
    runs = {}
    for first, second in (("native", "native"), ("native", "autograd"), ("autograd", "native")):
        p1 = fresh()
        upd1 = NativeDDPGfDUpdate(p1).train_on_batch if first == "native" else p1.train_on_batch
        for b in bs[:6]:
            upd1(*b)
        name = str(tmp_path / f"ck_{first}_{second}_{hidden[0]}")
        p1.save(name)
        sd = torch.load(name + "_critic_optimizer", weights_only=True)
        assert len(sd["state"]) == 6 and all(float(v["step"]) == 6 for v in sd["state"].values())
        assert all(v["exp_avg_sq"].abs().sum().item() > 0 for v in sd["state"].values())
        p2 = fresh()
        # the targets are not part of the reference's checkpoint (DDPGfD.py:371-382): carry them over by hand
        for k in ("actor_target", "critic_target"):
            p2._flat_params[k].copy_(p1._flat_params[k])
        nat2 = NativeDDPGfDUpdate(p2) if second == "native" else None
        p2.load(name)
        p2.total_it = 6
        if nat2 is not None:
            assert int(nat2.it.item()) == 6
        upd2 = nat2.train_on_batch if nat2 is not None else p2.train_on_batch
        if nat2 is None:
            p2._it_dev.fill_(6)
        for b in bs[6:]:
            upd2(*b)
        runs[(first, second)] = p2
    for key, pol in runs.items():
        for name in ("critic", "actor", "critic_target", "actor_target"):
            err = (pol._flat_params[name] - ref._flat_params[name]).abs().max().item()
            lr = 1e-3 if name.startswith("critic") else 1e-4
            assert err <= 0.1 * lr * 12, (key, name, err)
    # and a checkpoint WITHOUT the Adam state restarts the moments: visibly different weights (the bug this guards against)
    p3 = fresh()
    n3 = NativeDDPGfDUpdate(p3)
    p3.actor.load_state_dict(runs[("native", "native")].actor.state_dict())
    assert int(n3.it.item()) == 0


def test_capture_warmup_does_not_train():
    """GraphedTrainer.capture() on an EMPTY replay (the start of training): its eager warm-up steps run learner updates
    on all-padding batches; weights, Adam moments and counters must come out untouched, and nothing may be NaN."""
    from kinovagrasping_amd.ddpgfd import DDPGfD
    from kinovagrasping_amd.pipeline import GraphedTrainer
    from kinovagrasping_amd.replay import DeviceEpisodeReplay
    from kinovagrasping_amd.rollout import RolloutEngine
    from kinovagrasping_amd.sim import KinovaSim
    n = 128
    q0, hq = scenarios.config2_states(n)
    sim = KinovaSim(n, "CubeS", horizon=30, auto_reset=True)
    obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
    torch.manual_seed(2)
    policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=sim.device)
    before = {k: v.clone() for k, v in policy._flat_params.items()}
    replay = DeviceEpisodeReplay(n, capacity=512, horizon=30, device=sim.device)
    eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
    eng.start(obs0)
    tr = GraphedTrainer(sim, policy, replay, eng, batch_episodes=16)
    tr.capture()
    torch.cuda.synchronize()
    for k, v in before.items():
        assert torch.equal(policy._flat_params[k], v), k
    nat = tr.native
    assert int(nat.it.item()) == 0 and int(nat.it_head.item()) == 0
    for net in (nat.actor, nat.critic):
        assert net.exp_avg.abs().max().item() == 0 and net.exp_avg_sq.abs().max().item() == 0
    # the sampler on a ring with < 2 episodes: weight 0 everywhere, finite losses
    st, ac, ns, rw, nd, w = replay.sample_batch_nstep(16)
    assert w.sum().item() == 0
    assert torch.isfinite(nat.losses).all()
    # ... and training proper then works from the untouched state
    for _ in range(45):
        tr.step()
    tr.flush(finish_update=True)
    torch.cuda.synchronize()
    assert tr.updates >= 10 and int(nat.it.item()) == tr.updates and int(nat.it_head.item()) == 0
    assert all(torch.isfinite(v).all() for v in policy._flat_params.values())
    assert (policy._flat_params["actor"] - before["actor"]).abs().max().item() > 0
    sim.close()


def _philox4x32(c, k):
    """Philox4x32-10 on numpy uint32 arrays: counter c [..., 4], key k [2] -> [..., 4] (Salmon et al. 2011)"""
    c = [c[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(k[0]), np.uint64(k[1])
    M0, M1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c = [(hi1 ^ c[1] ^ k0) & mask, lo1, (hi0 ^ c[3] ^ k1) & mask, lo0]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
    return np.stack(c, -1).astype(np.uint32)


@pytest.mark.gpu
def test_window_sampler_with_in_kernel_uniforms_equals_the_sampler_fed_the_same_uniforms():
    """kr_sample_windows_draw draws its uniforms in the kernel (Philox4x32-10 keyed by the seed at counter (draw, index, tag));
    the same numbers computed on the host and handed to kr_sample_windows must select the same windows, and the extra
    [2R, S] block is next_state[:, 0] / next_state[:, -1].  Two calls with different `draw` differ, the same `draw` repeats."""
    from kinovagrasping_amd.replay import DeviceEpisodeReplay
    dev = torch.device("cuda", 0)
    B, H, n, S = 16, 30, 5, 82
    W = H - n
    rb = DeviceEpisodeReplay(8, capacity=40, horizon=H, device=dev)
    g = torch.Generator(device="cpu").manual_seed(3)
    K = 23
    rb.ep_state[:K] = torch.rand(K, H, S, generator=g).to(dev)
    rb.ep_next[:K] = torch.rand(K, H, S, generator=g).to(dev)
    rb.ep_action[:K] = torch.rand(K, H, 4, generator=g).to(dev)
    rb.ep_reward[:K] = torch.rand(K, H, generator=g).to(dev)
    rb.ep_not_done[:K] = 1.0
    rb.ep_len[:K] = torch.randint(7, H + 1, (K,), generator=g).to(dev)
    rb._count.fill_(K); rb._head.fill_(K)
    seed, draw_v = 0x1234567890ABCDEF, 77
    draw = torch.tensor([draw_v], dtype=torch.long, device=dev)
    out = rb.sample_batch_nstep(B, draw=draw, seed=seed)
    assert len(out) == 7
    key = np.array([seed & 0xFFFFFFFF, seed >> 32], dtype=np.uint64)

    def uniforms(idx, tag):
        c = np.stack([idx.astype(np.uint32), np.full_like(idx, draw_v & 0xFFFFFFFF, dtype=np.uint32),
                      np.full_like(idx, draw_v >> 32, dtype=np.uint32), np.full_like(idx, tag, dtype=np.uint32)], -1)
        return (_philox4x32(c, key)[..., 0] >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)

    u = np.concatenate([uniforms(np.arange(B), 0x5a4d), uniforms(np.arange(B * W), 0x5a4e)])
    ref = rb.sample_batch_nstep(B, uniforms=torch.as_tensor(u).to(dev))
    for a, b in zip(out[:6], ref):
        assert torch.equal(a, b)
    assert out[5].sum() > 0.5 * B * W * 0                                     # (weights exist; padding rows depend on the lengths)
    assert torch.equal(out[6][:B * W], out[2][:, 0]) and torch.equal(out[6][B * W:], out[2][:, -1])
    again = rb.sample_batch_nstep(B, draw=draw, seed=seed)
    assert all(torch.equal(a, b) for a, b in zip(out, again))
    other = rb.sample_batch_nstep(B, draw=draw + 1, seed=seed)
    assert not torch.equal(out[0], other[0])
    # the in-kernel uniforms are uniform: episode picks cover the ring's sampleable range
    eps = set()
    for d in range(40):
        o = rb.sample_batch_nstep(B, draw=draw + 2 + d, seed=seed)
        eps |= {int(x) for x in (o[0][::W, 0, 0].unsqueeze(1) == rb.ep_state[:K, :, 0].reshape(1, -1)).float().argmax(1) // H}
    assert len(eps) >= K - 3 and (K - 1) not in eps                            # the newest episode is never sampled


def _fill_ring(rb, K, seed, dev, offset):
    g = torch.Generator(device="cpu").manual_seed(seed)
    H, S = rb.horizon, rb.ep_state.shape[2]
    rb.ep_state[:K] = (torch.rand(K, H, S, generator=g) + offset).to(dev)        # `offset` tags the ring a row came from
    rb.ep_next[:K] = (torch.rand(K, H, S, generator=g) + offset).to(dev)
    rb.ep_action[:K] = torch.rand(K, H, 4, generator=g).to(dev)
    rb.ep_reward[:K] = torch.rand(K, H, generator=g).to(dev)
    rb.ep_not_done[:K] = 1.0
    rb.ep_len[:K] = torch.randint(7, H + 1, (K,), generator=g).to(dev)
    rb._count.fill_(K); rb._head.fill_(K % rb.capacity)


def test_expert_mix_sampler_splits_70_30_and_equals_the_torch_sampler_fed_the_same_uniforms():
    """DDPGfD.train_batch's batch (DDPGfD.py:232-254) as ONE launch (kr_sample_windows_mixed): with batch_size 64 and prob 0.3 the
    first int(64 * 0.7) = 44 episodes come from the agent ring, the other 20 from the expert ring, agent first; bit-equal to the
    concatenation of two torch-path sample_batch_nstep calls fed the same uniforms; and the in-kernel Philox draw equals the
    kernel fed those uniforms from the host."""
    from kinovagrasping_amd.replay import DeviceEpisodeReplay
    dev = torch.device("cuda", 0)
    B, H, n = 64, 30, 5
    W = H - n
    agent = DeviceEpisodeReplay(8, capacity=64, horizon=H, device=dev)
    expert = DeviceEpisodeReplay(8, capacity=32, horizon=H, device=dev)
    _fill_ring(agent, 41, 5, dev, 0.0)
    _fill_ring(expert, 19, 6, dev, 10.0)
    g = torch.Generator(device="cpu").manual_seed(9)
    u = torch.rand(B * (W + 1), generator=g).to(dev)
    out = agent.sample_mixed(expert, B, 0.3, uniforms=u)
    b_agent = int(B * (1 - 0.3))
    assert b_agent == 44
    from_expert = (out[0][:, 0, 0] >= 10.0).view(B, W)
    assert not from_expert[:b_agent].any() and from_expert[b_agent:].all()           # 44 agent episodes, then 20 expert episodes
    # the checker: the torch arithmetic of DeviceEpisodeReplay (native kernels off), two independent samplers, concatenated
    agent.native = expert.native = False
    ref = agent.sample_mixed(expert, B, 0.3, uniforms=u)
    agent.native = expert.native = True
    w = out[5] > 0
    assert torch.equal(out[5], ref[5]) and w.sum() > 0.5 * B * W
    for a, b in zip(out[:5], ref[:5]):
        assert torch.equal(a[w], b[w])                                               # padding rows (weight 0) are don't-care
    # in-kernel uniforms (the graphed trainer's form): same Philox stream as the single-ring sampler
    seed, draw_v = 0xA5A5A5A55A5A5A5A, 123
    draw = torch.tensor([draw_v], dtype=torch.long, device=dev)
    drawn = agent.sample_mixed(expert, B, 0.3, draw=draw, seed=seed)
    assert len(drawn) == 7
    key = np.array([seed & 0xFFFFFFFF, seed >> 32], dtype=np.uint64)

    def uniforms(idx, tag):
        c = np.stack([idx.astype(np.uint32), np.full_like(idx, draw_v & 0xFFFFFFFF, dtype=np.uint32),
                      np.full_like(idx, draw_v >> 32, dtype=np.uint32), np.full_like(idx, tag, dtype=np.uint32)], -1)
        return (_philox4x32(c, key)[..., 0] >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)

    uh = torch.as_tensor(np.concatenate([uniforms(np.arange(B), 0x5a4d), uniforms(np.arange(B * W), 0x5a4e)])).to(dev)
    fed = agent.sample_mixed(expert, B, 0.3, uniforms=uh)
    assert all(torch.equal(a, b) for a, b in zip(drawn[:6], fed))
    assert torch.equal(drawn[6][:B * W], drawn[2][:, 0]) and torch.equal(drawn[6][B * W:], drawn[2][:, -1])
    # prob 0: every episode from the agent ring, and equal to the single-ring kernel
    only = agent.sample_mixed(expert, B, 0.0, uniforms=u)
    single = agent.sample_batch_nstep(B, uniforms=u)
    assert all(torch.equal(a, b) for a, b in zip(only, single))


def test_graphed_trainer_with_the_expert_mix_trains_on_demonstrations():
    """DDPGfD on the product path: demonstrations (combined controller, expert_data.py loop) fill an expert ring, GraphedTrainer
    samples 70 % agent + 30 % expert episodes inside its captured update (one launch), trains, stays finite; the expert rows of
    the captured batch really are demonstration transitions (their actions are the controllers': fingers in [0.5, 0.8] or the
    lift action), while agent rows carry exploration noise."""
    from kinovagrasping_amd.ddpgfd import DDPGfD
    from kinovagrasping_amd.demonstrators import run_controller_episodes
    from kinovagrasping_amd.pipeline import GraphedTrainer
    from kinovagrasping_amd.replay import DeviceEpisodeReplay
    from kinovagrasping_amd.rollout import RolloutEngine
    from kinovagrasping_amd.sim import KinovaSim
    n = 256
    q0, hq = scenarios.config2_states(n)
    sim = KinovaSim(n, "CubeS", horizon=30, auto_reset=True)
    reset = lambda: sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
    expert = DeviceEpisodeReplay(n, capacity=n, horizon=30, device=sim.device)
    demo = run_controller_episodes(sim, reset().clone(), expert, mode="combined")
    assert expert.count >= 0.9 * n and demo["success"].float().mean() > 0.5      # (0.62: the recorded naive-controller map has 60 % success cells)
    obs0 = reset()
    torch.manual_seed(2)
    policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=sim.device)
    replay = DeviceEpisodeReplay(n, capacity=1024, horizon=30, device=sim.device)
    eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
    eng.start(obs0)
    tr = GraphedTrainer(sim, policy, replay, eng, batch_episodes=16, expert_replay=expert, expert_prob=0.3)
    tr.capture()
    w0 = policy._flat_params["critic"].clone()
    for _ in range(70):
        tr.step()
    tr.flush()
    torch.cuda.synchronize()
    assert tr.updates >= 35
    st, ac, ns, rw, nd, w = tr.batch[:6]
    W, b_agent = 25, int(16 * 0.7)
    assert b_agent == 11 and st.shape[0] == 16 * W
    live = w.view(16, W) > 0
    assert live[:b_agent].any() and live[b_agent:].any()
    exp_actions = ac.view(16, W, 5, 4)[b_agent:][live[b_agent:]]
    fingers = exp_actions[..., 1:]
    assert ((fingers >= 0.5 - 1e-6) & (fingers <= 0.8 + 1e-6)).all()                  # check_vel_in_range / naive / lift velocities
    assert set(exp_actions[..., 0].unique().tolist()) <= {0.0, 0.6000000238418579}  # wrist: 0 or wrist_lift_velocity
    agent_fingers = ac.view(16, W, 5, 4)[:b_agent][live[:b_agent]][..., 1:]
    assert ((agent_fingers < 0.5 - 1e-3) | (agent_fingers > 0.8 + 1e-3)).any()       # the actor + noise leaves that band
    assert torch.isfinite(tr.native.losses).all() and torch.isfinite(policy._flat_params["actor"]).all()
    assert (policy._flat_params["critic"] - w0).abs().max().item() > 0
    sim.close()


def test_reference_goldens_through_the_device_code_paths(golden_dir, tmp_path):
    """The section-8f goldens that touch device code, run ON the GPU (the CPU suite runs the same arithmetic on CPU tensors):
    (a) demonstrators.controller_action on cuda tensors against the 600 reference answers of tests/golden/controllers.npz;
    (b) the reference-written replay bundle loaded into the DEVICE ring (kr_* kernels), sampled with the native window kernel -
        every sampled window is a window of the reference's own episodes - and saved back to identical episode content."""
    from kinovagrasping_amd.demonstrators import controller_action
    from kinovagrasping_amd.replay import DeviceEpisodeReplay, load_reference_bundle
    dev = torch.device("cuda", 0)
    g = np.load(golden_dir / "controllers.npz")
    n = len(g["obs21"])
    obs = torch.zeros(n, 82, dtype=torch.float64, device=dev)
    for col, key in ((21, "obs21"), (78, "obs78"), (79, "obs79"), (81, "obs81")):
        obs[:, col] = torch.as_tensor(g[key]).to(dev)
    init_x, init_dot, lift = (torch.as_tensor(g[k]).to(dev) for k in ("init21", "init81", "lift"))
    for mode, key in (("naive", "action_naive"), ("position-dependent", "action_position_dependent"), ("combined", "action_combined")):
        a = controller_action(mode, obs, init_x, init_dot, lift).cpu().numpy()
        assert np.abs(a - g[key]).max() < 1e-12, mode
    exp = np.load(golden_dir / "replay_bundle_expected.npz")
    ring = DeviceEpisodeReplay(n_envs=2, capacity=8, horizon=30, device=dev)
    assert ring.native
    info = ring.load(golden_dir / "replay_bundle")
    assert info.tolist() == [100, 49, 3, 3] and ring.count == 3 and ring.ep_len[:3].tolist() == [7, 30, 12]
    u = torch.rand(16 * 26, generator=torch.Generator(device="cpu").manual_seed(4)).to(dev)
    st, ac, ns, rw, nd, w = ring.sample_batch_nstep(16, uniforms=u)
    lens = [7, 30]                                  # the newest episode (12 steps) is never sampled (utils.py:259)
    rows = {i: exp[f"ep{i}_state"] for i in range(2)}
    st_h, w_h = st.cpu().numpy(), w.cpu().numpy()
    assert w_h.sum() > 16
    for r in np.flatnonzero(w_h > 0):
        hit = False
        for i, L in enumerate(lens):
            for s0 in range(L - 5 + 1):
                if np.array_equal(st_h[r], rows[i][s0:s0 + 5].astype(np.float32)):
                    hit = True
        assert hit, r
    ring.save(tmp_path / "d")
    eps, _ = load_reference_bundle(tmp_path / "d")
    for i, e in enumerate(eps):
        for k, v in e.items():
            np.testing.assert_allclose(v, exp[f"ep{i}_{k}"], rtol=0, atol=1e-6)


def test_forked_update_equals_the_single_stream_update_bit_for_bit(monkeypatch):
    """Round 5: the LDS-free update's independent launches run on two side streams that leave from and rejoin the phase's stream
    (learner_native: _branch / _join; opt-in, KS_LEARNER_FORK=1).  Same kernels, same inputs, deterministic reductions: after 8 updates - eager and replayed from a
    captured graph - every parameter, gradient and Adam moment equals the single-stream run's (KS_LEARNER_FORK=0) bit for bit."""
    from kinovagrasping_amd.ddpgfd import DDPGfD
    from kinovagrasping_amd.learner_native import NativeDDPGfDUpdate
    dev = torch.device("cuda", 0)
    bs = _batches(dev, 8)

    def run(fork, graph):
        monkeypatch.setenv("KS_LEARNER_FORK", "1" if fork else "0")
        torch.manual_seed(7)
        pol = DDPGfD(82, 4, 0.8, 5, hidden=(256, 256), device=dev)
        nat = NativeDDPGfDUpdate(pol)
        assert nat.lds_free and nat.fork == fork
        if not graph:
            for b in bs:
                nat.train_on_batch(*b)
        else:
            static = [t.clone() for t in bs[0]]
            s = torch.cuda.Stream(dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                nat.train_on_batch(*static)                                   # warm-up outside the capture (allocations, library handles)
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize()
            torch.manual_seed(7)
            pol2 = DDPGfD(82, 4, 0.8, 5, hidden=(256, 256), device=dev)
            for k in pol._flat_params:
                pol._flat_params[k].copy_(pol2._flat_params[k])
            for net in (nat.actor, nat.critic):
                net.grad.zero_(); net.exp_avg.zero_(); net.exp_avg_sq.zero_()
            nat.it.zero_(); nat.it_head.zero_(); pol.total_it = 0
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                nat.train_on_batch(*static)
            for b in bs:
                for dst, src in zip(static, b):
                    dst.copy_(src)
                g.replay()
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in pol._flat_params.items()}, [(n.grad.clone(), n.exp_avg.clone(), n.exp_avg_sq.clone()) for n in (nat.actor, nat.critic)]

    ref_p, ref_o = run(False, False)
    for fork, graph in ((True, False), (True, True), (False, True)):
        p, o = run(fork, graph)
        for k in ref_p:
            assert torch.equal(p[k], ref_p[k]), (fork, graph, k, (p[k] - ref_p[k]).abs().max().item())
        for a, b in zip(o, ref_o):
            assert all(torch.equal(x, y) for x, y in zip(a, b)), (fork, graph)
