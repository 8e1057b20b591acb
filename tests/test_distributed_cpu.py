"""world_size-2 gloo test of the one exchange step of the path: gradient averaging across
env shards.  Two ranks train on disjoint half batches; with the flat all-reduce their replicas stay
bit-identical and equal a single process that trains on the concatenated batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kinovagrasping_amd.ddpgfd import DDPGfD


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _make_batch(seed, rows):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(rows, 5, 82, generator=g), torch.rand(rows, 5, 4, generator=g) * 0.8,
            torch.randn(rows, 5, 82, generator=g), torch.rand(rows, 5, generator=g))


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(123)                      # identical initial replicas
    pol = DDPGfD(82, 4, 0.8, 5, hidden=(64, 48))
    for it in range(12):                        # crosses the soft target update at call 10
        full = _make_batch(100 + it, 32)
        half = tuple(x[rank * 16:(rank + 1) * 16] for x in full)
        pol.train_on_batch(*half)
    torch.save({k: v for k, v in pol.actor.state_dict().items()} | {"c." + k: v for k, v in pol.critic.state_dict().items()}
               | {"t." + k: v for k, v in pol.actor_target.state_dict().items()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k          # replicas stay bit-identical
    torch.manual_seed(123)
    pol = DDPGfD(82, 4, 0.8, 5, hidden=(64, 48))
    for it in range(12):
        pol.train_on_batch(*_make_batch(100 + it, 32))
    ref = {k: v for k, v in pol.actor.state_dict().items()} | {"c." + k: v for k, v in pol.critic.state_dict().items()} \
        | {"t." + k: v for k, v in pol.actor_target.state_dict().items()}
    for k in ref:
        torch.testing.assert_close(r0[k], ref[k], rtol=1e-4, atol=1e-6)


def test_env_sharding_is_independent_of_world_size():
    """rank r of G owns global envs [r*n, (r+1)*n): start states and action streams depend on the global
    env id only (scenarios.config2_states / config_actions), so the union over ranks is the same for any G."""
    from kinovagrasping_amd import scenarios
    q_all, _ = scenarios.config2_states(8)
    a_all = scenarios.config_actions(8, 3, base_seed=1000)
    for world in (1, 2, 4):
        n = 8 // world
        q = np.concatenate([scenarios.config2_states(8)[0][:, r * n:(r + 1) * n] for r in range(world)], 1)
        a = np.concatenate([scenarios.config_actions(n, 3, base_seed=1000 + r * n) for r in range(world)], 2)
        np.testing.assert_array_equal(q, q_all)
        np.testing.assert_array_equal(a, a_all)


def test_config5_cohort_draw_gives_whole_groups_and_keeps_the_iid_stream():
    """scenarios.config5_states(cohort=16): every shape's env count is a multiple of the stepping kernel's 16-env groups (N / 16 groups, which the
    free-running rollout deals evenly to its persistent workgroups), each env's shape is still uniform over the 14, and orientation / start rows /
    mass / friction are the per-env draws of cohort = 1 (same generator stream); a rank's shard is a slice of the global draw."""
    import numpy as np
    from kinovagrasping_amd import scenarios
    n = 8192
    o1, names1, q1, hq1, mf1 = scenarios.config5_states(n, seed=5)
    o16, names16, q16, hq16, mf16 = scenarios.config5_states(n, seed=5, cohort=16)
    cnt = np.bincount(o16, minlength=14)
    assert (cnt % 16 == 0).all() and cnt.sum() == n and int(((cnt + 15) // 16).sum()) == n // 16
    assert (o16.reshape(-1, 16) == o16.reshape(-1, 16)[:, :1]).all() and (o16[::16] == o1[:n // 16]).all()
    assert names16 == names1 and np.array_equal(mf16, mf1) and np.array_equal(hq16, hq1)
    assert cnt.min() >= 0.5 * n / 14 and cnt.max() <= 1.6 * n / 14            # uniform over the shapes (512 cohort draws)
    # cohorts of a 2-rank run: every rank's shard [r n, (r + 1) n) keeps whole groups too
    o2 = scenarios.config5_states(2 * 4096, seed=5, cohort=16)[0]
    for r in range(2):
        assert (np.bincount(o2[r * 4096:(r + 1) * 4096], minlength=14) % 16 == 0).all()
