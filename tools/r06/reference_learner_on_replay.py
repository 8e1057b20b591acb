#!/usr/bin/env python3
"""CPU, this container only (imports the reference): does the REFERENCE's own learner run away like ours?  (VERDICT r5 weak #10 / next #7.)

examples/train_ddpgfd.py in the reference's schedule (107 updates per env-step of 32 envs, tau 0.0005 every 10th update, 30 % expert episodes) learns to
0.92 lift success and then diverges: the critic's loss grows throughout (22 -> 5e4 -> 4e7) until the policy collapses.  The update is pinned to the
reference's to 1e-6 per call (tests/test_learner_golden.py), which makes a bug of ours unlikely but does not show that DDPGfD.py behaves the same over
100 k updates.  Here both learners continue FROM THE SAME STATE on THE SAME REPLAY: the replay and the four networks dumped from a GPU run
(examples/train_ddpgfd.py --dump-replay) are loaded into the reference's ReplayBuffer_Queue + DDPGfD and into this repo's HostEpisodeReplay + DDPGfD
(autograd implementation, CPU), and each runs N more updates on the now static replay with its own np.random stream.  The reference's train_batch is
given THIS repo's HostEpisodeReplay as its buffers: its own sampler builds tensors from lists of arrays (~1 s per batch of 1600 windows); HostEpisodeReplay draws
the same windows from the same np.random stream bit for bit (tests/test_learner_golden.py::test_sampler_consumes_the_same_random_stream).  The update rule -
targets, n-step returns, losses, optimiser steps, target schedule - is the reference's code.
usage: python tools/r06/reference_learner_on_replay.py gpurun_out/r06/ref_schedule_replay.npz [updates]"""
import sys, time
from pathlib import Path
import numpy as np
import torch
REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
sys.path.insert(0, "/root/reference/gym-kinova-gripper")
import DDPGfD as ref_ddpg          # noqa: E402
import utils as ref_utils          # noqa: E402
from kinovagrasping_amd.ddpgfd import DDPGfD      # noqa: E402
from kinovagrasping_amd.replay import HostEpisodeReplay      # noqa: E402

D = dict(np.load(sys.argv[1]))          # (every array decompressed once)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
torch.set_num_threads(4)


def episodes(name):
    lens, off = D[f"{name}_lens"], 0
    for L in lens:
        yield {f: D[f"{name}_{f}"][off:off + L] for f in ("state", "action", "next_state", "reward")}
        off += L


def load_nets(pol):
    for name in ("actor", "critic", "actor_target", "critic_target"):
        net = getattr(pol, name)
        net.load_state_dict({k: torch.from_numpy(D[f"{name}.{k}"]) for k in net.state_dict()})


# the reference's classes fix the hidden widths at 400-300 (DDPGfD.py:15-50); their forward() only uses self.l1 / l2 / l3, so the layers are re-made
# at the dump's widths (the GPU run's, 256-256 by default) - update rule, losses, sampling and target schedule stay the reference's code
h1, h2 = D["actor.l1.weight"].shape[0], D["actor.l2.weight"].shape[0]
ref = ref_ddpg.DDPGfD(82, 4, 0.8, 5, batch_size=64)
ref.device = torch.device("cpu")
for n_ in ("actor", "critic", "actor_target", "critic_target"):
    net = getattr(ref, n_).cpu()
    net.l1 = torch.nn.Linear(82 + (4 if "critic" in n_ else 0), h1)
    net.l2 = torch.nn.Linear(h1, h2)
    net.l3 = torch.nn.Linear(h2, 4 if "actor" in n_ else 1)
    setattr(ref, n_, net)
load_nets(ref)
ref.actor_optimizer = torch.optim.Adam(ref.actor.parameters(), lr=1e-4)
ref.critic_optimizer = torch.optim.Adam(ref.critic.parameters())
ref_utils.device = ref_ddpg.device = torch.device("cpu")
ours_b = {k: HostEpisodeReplay() for k in ("agent", "expert")}
for k in ours_b:
    for ep in episodes(k):
        L = len(ep["reward"])
        nd = np.ones(L, np.float32); nd[-1] = 0
        ours_b[k].add_episode_arrays(ep["state"], ep["action"], ep["next_state"], ep["reward"], nd)
ours = DDPGfD(82, 4, 0.8, 5, tau=0.0005, batch_size=64, hidden=(h1, h2))
load_nets(ours)
print(f"replay: agent {ours_b['agent'].replay_ep_num} episodes, expert {ours_b['expert'].replay_ep_num}; networks after {int(D['updates'])} updates of the GPU run; {N} more updates each "
      f"on the static replay (64 episodes x 25 five-step windows per update, 30 % expert, targets every 10th update at tau 0.0005)")
sys.stdout.flush()
print(f"{'updates':>8} {'reference critic loss':>22} {'ours critic loss':>18}   {'reference |Q| mean':>18} {'ours |Q| mean':>14}")
st_r, st_o = np.random.RandomState(1).get_state(), np.random.RandomState(1).get_state()
acc_r, acc_o, t0 = [], [], time.time()
probe = torch.from_numpy(np.concatenate([e["state"] for _, e in zip(range(50), episodes("agent"))]))
for it in range(1, N + 1):
    np.random.set_state(st_r)
    lr_ = ref.train_batch(30, ours_b["expert"], ours_b["agent"], 5, prob=0.3)
    st_r = np.random.get_state()
    np.random.set_state(st_o)
    lo_ = ours.train_batch(30, ours_b["expert"], ours_b["agent"], 5, prob=0.3)
    st_o = np.random.get_state()
    acc_r.append(float(lr_[1])); acc_o.append(float(lo_[1]))            # (actor loss, critic loss, critic L1, critic LN) - DDPGfD.py:367
    if it % 500 == 0:
        with torch.no_grad():
            qr = ref.critic(probe, ref.actor(probe)).abs().mean().item()
            qo = ours.critic(probe, ours.actor(probe)).abs().mean().item()
        print(f"{it:8d} {np.mean(acc_r):22.3f} {np.mean(acc_o):18.3f}   {qr:18.2f} {qo:14.2f}   ({time.time() - t0:.0f} s)", flush=True)
        acc_r, acc_o = [], []
