#!/usr/bin/env python3
"""End-to-end DDPGfD training on the batched simulator (the counterpart of
`python main_DDPGfD.py --mode train ...` in the reference, gym-kinova-gripper/main_DDPGfD.py:333-537):

  1. expert data: the naive demonstrator (expert_data.py:596-607) on start positions sampled from the
     no_noise start table fills the expert replay;
  2. training: N envs roll out clip(pi(s) + N(0, 0.08), 0, 0.8) with the scripted lift after check_grasp,
     agent transitions go to the device replay, and every env-step one DDPGfD update mixes 70 % agent /
     30 % expert episodes (DDPGfD.py:232-254);
  3. periodic evaluation without exploration noise: lift success rate.

    python examples/train_ddpgfd.py --envs 1024 --steps 600 --hidden 400 300
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from kinovagrasping_amd import scenarios                       # noqa: E402
from kinovagrasping_amd.ddpgfd import DDPGfD                   # noqa: E402
from kinovagrasping_amd.demonstrators import run_controller_episodes, run_naive_episodes  # noqa: E402
from kinovagrasping_amd.replay import DeviceEpisodeReplay      # noqa: E402
from kinovagrasping_amd.rollout import RolloutEngine           # noqa: E402
from kinovagrasping_amd.sim import KinovaSim                   # noqa: E402


def start_states(n, shape, rng):
    tab = scenarios.start_coord_table(shape)
    q = np.zeros((16, n)); q[12] = 1.0
    q[9:12] = tab[rng.randint(0, len(tab), n)].T
    return q, np.repeat(scenarios.hand_quat_for("normal")[:, None], n, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--shape", default="CubeS")
    ap.add_argument("--hidden", type=int, nargs=2, default=[400, 300])
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--expert-episodes", type=int, default=2048)
    args = ap.parse_args()
    torch.manual_seed(args.seed)
    rng = np.random.RandomState(args.seed)
    dev = torch.device("cuda", 0)
    n = args.envs

    # 1. expert replay from the naive demonstrator
    expert = DeviceEpisodeReplay(n, capacity=args.expert_episodes, device=dev)
    sim = KinovaSim(n, args.shape, auto_reset=False, horizon=30)
    succ = []
    while expert.count < args.expert_episodes:
        q0, hq = start_states(n, args.shape, rng)
        out = run_naive_episodes(sim, sim.reset(torch.as_tensor(q0), torch.as_tensor(hq)), expert)
        succ.append(out["success"].float().mean().item())
    print(f"expert replay: {expert.count} episodes, naive-controller lift success {np.mean(succ):.2f}")
    sim.close()

    # 2. training
    sim = KinovaSim(n, args.shape, auto_reset=True, horizon=30)
    q0, hq = start_states(n, args.shape, rng)
    policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=tuple(args.hidden), device=dev)
    agent = DeviceEpisodeReplay(n, capacity=max(4 * n, 4096), device=dev)
    eng = RolloutEngine(sim, policy, agent)
    eng.start(sim.reset(torch.as_tensor(q0), torch.as_tensor(hq)))
    lifted = episodes = 0
    t0 = time.perf_counter()
    for it in range(args.steps):
        reward, done = eng.step()
        lifted += int(((reward > 0) & done).sum())
        episodes += int(done.sum())
        if agent.count >= 2:
            ag = agent.sample_batch_nstep(int(64 * 0.7))
            ex = expert.sample_batch_nstep(64 - int(64 * 0.7))
            batch = [torch.cat((a, e), 0) for a, e in zip(ag, ex)]
            losses = policy.train_on_batch(batch[0], batch[1], batch[2], batch[3], batch[5])
        if (it + 1) % 60 == 0:
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"step {it + 1:5d}  episodes {episodes:7d}  lift rate {lifted / max(1, episodes):.3f}  "
                  f"actor loss {losses[0].item():8.3f}  critic loss {losses[1].item():9.3f}  {n * (it + 1) / dt:9.0f} env-steps/s")
            lifted = episodes = 0
    sim.close()


if __name__ == "__main__":
    main()
