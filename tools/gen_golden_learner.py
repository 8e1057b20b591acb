#!/usr/bin/env python3
"""Dev-only: golden vectors for the learner path from the REFERENCE's own DDPGfD.py / utils.py
(importable as-is: only torch + numpy).  Writes tests/golden/learner.npz.

Captured: initial actor/critic state_dicts (seeded), a replay buffer of synthetic episodes, the batch
that ReplayBuffer_Queue.sample_batch_nstep returns under a seeded np.random (with the exact index
draws), the four losses of train_batch and the parameters after 1 and after 10 calls (the 10th runs
the soft target update).  CPU, fp32.
"""
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, "/root/reference/gym-kinova-gripper")
import DDPGfD as ref_ddpg  # noqa: E402
import utils as ref_utils  # noqa: E402


def flat(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def main():
    torch.manual_seed(2)
    np.random.seed(2)
    state_dim, action_dim, max_action, n = 82, 4, 0.8, 5
    pol = ref_ddpg.DDPGfD(state_dim, action_dim, max_action, n, batch_size=6)
    out = {}
    for name, net in (("actor", pol.actor), ("critic", pol.critic)):
        for k, v in flat(net.state_dict()).items():
            out[f"init_{name}.{k}"] = v
    # synthetic episodes of different lengths (a done flag ends each)
    rng = np.random.RandomState(7)
    lens = [30, 30, 12, 30, 21, 30, 9, 30, 30]
    ep_state, ep_action, ep_next, ep_reward = [], [], [], []
    buf_a = ref_utils.ReplayBuffer_Queue(state_dim, action_dim, 10000, n)
    buf_e = ref_utils.ReplayBuffer_Queue(state_dim, action_dim, 10000, n)
    for bi, buf in enumerate((buf_a, buf_e)):
        for L in lens:
            s = rng.uniform(-1, 1, (L + 1, state_dim)).astype(np.float32)
            a = rng.uniform(0, 0.8, (L, action_dim)).astype(np.float32)
            r = np.zeros(L, dtype=np.float32)
            if rng.rand() < 0.5:
                r[-1] = 50.0
            buf.add_episode(1)
            for t in range(L):
                buf.add(s[t], a[t], s[t + 1], r[t], float(t == L - 1))
            buf.add_episode(0)
            ep_state.append(s); ep_action.append(a); ep_reward.append(r)
    out["ep_lens"] = np.array(lens * 2)
    out["ep_state"] = np.concatenate(ep_state)
    out["ep_action"] = np.concatenate(ep_action)
    out["ep_reward"] = np.concatenate(ep_reward)
    # sampler known answers: record the np.random draws by replaying the same seed
    np.random.seed(11)
    st, ac, ns, rw, nd = buf_a.sample_batch_nstep(4)
    out["samp_state"], out["samp_action"], out["samp_next"] = st.numpy(), ac.numpy(), ns.numpy()
    out["samp_reward"], out["samp_not_done"] = rw.numpy(), nd.numpy()
    # train_batch: agent 70% / expert 30% mix (DDPGfD.py:232-254); capture the exact batch per call
    captured = []
    orig = ref_utils.ReplayBuffer_Queue.sample_batch_nstep
    def wrapped(self, bs, num=5):
        res = orig(self, bs, num)
        captured.append([t.numpy().copy() for t in res])
        return res
    ref_utils.ReplayBuffer_Queue.sample_batch_nstep = wrapped
    np.random.seed(5)
    losses = []
    for it in range(10):
        losses.append(pol.train_batch(30, buf_e, buf_a, 5, prob=0.3))
        if it in (0, 9):
            for name, net in (("actor", pol.actor), ("critic", pol.critic), ("actor_target", pol.actor_target),
                              ("critic_target", pol.critic_target)):
                for k, v in flat(net.state_dict()).items():
                    # full tensors would make the fixture ~7 MB: keep exact sums and a strided sample
                    out[f"after{it + 1}_{name}.{k}.stats"] = np.array([v.astype(np.float64).sum(), np.abs(v.astype(np.float64)).sum()])
                    out[f"after{it + 1}_{name}.{k}.sample"] = v.ravel()[::max(1, v.size // 256)][:256].copy()
    out["losses"] = np.array(losses, dtype=np.float64)      # actor, critic, L1, LN
    # batches are reproducible from the episodes + the seeded np.random stream (tests re-sample them
    # with the repo's restatement of the sampler); keep call 0 as a direct known answer
    ag, ex = captured[0], captured[1]
    for j, nm in enumerate(("state", "action", "next", "reward", "not_done")):
        out[f"batch0_{nm}"] = np.concatenate([ag[j], ex[j]], 0)
    out["batch_rows"] = np.array([captured[2 * i][0].shape[0] + captured[2 * i + 1][0].shape[0] for i in range(10)])
    # select_action known answer
    s0 = rng.uniform(-1, 1, state_dim).astype(np.float32)
    out["sel_state"], out["sel_action"] = s0, pol.select_action(s0)
    dst = REPO / "tests" / "golden" / "learner.npz"
    np.savez_compressed(dst, **out)
    print("wrote", dst, len(out), "arrays; losses[0]", losses[0], "losses[9]", losses[9])
    print("batch rows per call", out["batch_rows"])


if __name__ == "__main__":
    main()
