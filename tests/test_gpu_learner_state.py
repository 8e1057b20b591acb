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
