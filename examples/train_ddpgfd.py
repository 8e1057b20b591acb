#!/usr/bin/env python3
"""End-to-end DDPGfD training on the batched simulator (the counterpart of
`python main_DDPGfD.py --mode train ...` in the reference, gym-kinova-gripper/main_DDPGfD.py:333-537):

  1. expert data: a scripted demonstrator (expert_data.py:596-671, loop :746-804) on start positions sampled from the
     no_noise start table fills the expert replay;
  2. training: N envs roll out clip(pi(s) + N(0, 0.08), 0, 0.8) with the scripted lift after check_grasp,
     agent transitions go to the device replay, and every env-step one DDPGfD update mixes 70 % agent /
     30 % expert episodes (DDPGfD.py:232-254);
  3. periodic evaluation without exploration noise (eval_policy, main_DDPGfD.py:130-272): lift success rate on 1024 fresh start positions.
     (The lift rate of the TRAINING rollouts is depressed by the exploration noise on the wrist channel: clip(pi + N(0, 0.08), 0, 0.8) lifts the
     hand by ~5 mm per env-step on average while the fingers close - as in the reference, main_DDPGfD.py:443-446.)

    python examples/train_ddpgfd.py --envs 1024 --steps 600 --hidden 256 256 [--free-running] [--expert-prob 0]

Measured curves: profiles/r04_training_curves.txt.  One update per env-step of 4096 envs (BASELINE config 3's workload) is 1e4 times fewer
updates per stored transition than the reference's 100 updates per episode of one env (main_DDPGfD.py:474-476); with the reference's target
rate (tau 0.0005 every 10th update, DDPGfD.py:64-66,360-366) the target networks move 26 % of the way in 6000 updates and nothing
propagates.  The defaults below therefore let the targets follow EVERY update at tau = 0.001 (--tau 0.0005 --target-every 10 are the
reference's values): with the 30 % expert mix the noise-free evaluation reaches 0.92 lift success after 600 updates and stays >= 0.80 for
9000 consecutive updates (the demonstrator itself: 0.64); plain DDPG needs 7000 updates to leave the 0.65 plateau of a constant closing action.
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
from kinovagrasping_amd.demonstrators import run_controller_episodes  # noqa: E402
from kinovagrasping_amd.evaluate import eval_policy            # noqa: E402
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
    ap.add_argument("--hidden", type=int, nargs=2, default=[256, 256])
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--expert-episodes", type=int, default=2048)
    ap.add_argument("--controller", default="combined", choices=["naive", "position-dependent", "combined"])
    ap.add_argument("--expert-prob", type=float, default=0.3, help="share of expert episodes in a batch (DDPGfD.py:232-254); 0: plain DDPG, no demonstrations")
    ap.add_argument("--eval-every", type=int, default=600, help="env-steps between evaluations without exploration noise (0: none)")
    ap.add_argument("--free-running", action="store_true", help="the persistent rollout kernel (ks_rollout) instead of one launch per env-step")
    ap.add_argument("--batch-episodes", type=int, default=64, help="episodes per update (x 25 five-step windows each); the reference: 64")
    ap.add_argument("--updates-per-step", type=int, default=1, help="learner updates per env-step of the whole batch of envs")
    ap.add_argument("--actor-lr", type=float, default=1e-4, help="reference: 1e-4 (DDPGfD.py:57)")
    ap.add_argument("--critic-lr", type=float, default=1e-3, help="reference: Adam's default 1e-3 (DDPGfD.py:61)")
    ap.add_argument("--tau", type=float, default=0.001, help="soft target update rate (reference: 0.0005, main_DDPGfD.py:894 - for 100 updates per episode of one env)")
    ap.add_argument("--target-every", type=int, default=1, help="target networks follow every this many updates (reference: 10, DDPGfD.py:64-66,360-366)")
    ap.add_argument("--expl-noise", type=float, default=0.1, help="exploration noise, std = 0.8 x this (main_DDPGfD.py:443-446: 0.1)")
    ap.add_argument("--dump-replay", default=None, help="write the replay (last 1500 agent + first 400 expert episodes) and the four networks as one .npz "
                    "(tools/r06/reference_learner_on_replay.py runs the REFERENCE's train_batch on it)")
    ap.add_argument("--save", default=None, help="write the trained policy as the reference's 4-file checkpoint with this prefix")
    args = ap.parse_args()
    torch.manual_seed(args.seed)
    rng = np.random.RandomState(args.seed)
    dev = torch.device("cuda", 0)
    n = args.envs

    # 1. expert replay: the combined controller with the demonstration loop of expert_data.py:746-804
    expert = DeviceEpisodeReplay(n, capacity=args.expert_episodes, device=dev) if args.expert_prob > 0 else None
    sim = KinovaSim(n, args.shape, auto_reset=False, horizon=30)
    succ = []
    while expert is not None and expert.count < args.expert_episodes:
        q0, hq = start_states(n, args.shape, rng)
        out = run_controller_episodes(sim, sim.reset(torch.as_tensor(q0), torch.as_tensor(hq)), expert, mode=args.controller)
        succ.append(out["success"].float().mean().item())
    if expert is not None:
        print(f"expert replay: {expert.count} episodes, {args.controller}-controller lift success {np.mean(succ):.2f}")
    print(f"targets follow every {args.target_every} update(s) at tau {args.tau} (reference: every 10th at 0.0005), batch {args.batch_episodes} episodes, "
          f"{args.updates_per_step} update(s) per env-step, actor / critic lr {args.actor_lr} / {args.critic_lr}, expert share {args.expert_prob}")
    sim.close()

    # 2. training on the product path: HIP-graph trainer, native MFMA learner, every batch 44 agent + 20 expert episodes sampled by
    #    one launch inside the captured update (pipeline.GraphedTrainer; --free-running: the persistent rollout kernel, AsyncTrainer)
    from kinovagrasping_amd.pipeline import AsyncTrainer, GraphedTrainer
    sim = KinovaSim(n, args.shape, auto_reset=True, horizon=30)
    q0, hq = start_states(n, args.shape, rng)
    policy = DDPGfD(82, 4, 0.8, 5, tau=args.tau, batch_size=args.batch_episodes, hidden=tuple(args.hidden), device=dev)
    policy.network_repl_freq = args.target_every
    policy.actor_optimizer.param_groups[0]["lr"] = args.actor_lr
    policy.critic_optimizer.param_groups[0]["lr"] = args.critic_lr
    agent = DeviceEpisodeReplay(n, capacity=max(4 * n, 4096), device=dev)
    eng = RolloutEngine(sim, policy, agent, expl_noise=args.expl_noise)
    eng.start(sim.reset(torch.as_tensor(q0), torch.as_tensor(hq)))
    Trainer = AsyncTrainer if args.free_running else GraphedTrainer
    tr = Trainer(sim, policy, agent, eng, batch_episodes=args.batch_episodes, expert_replay=expert, expert_prob=args.expert_prob if expert is not None else 0.3,
                 updates_per_step=args.updates_per_step)
    tr.capture()
    if args.free_running:
        tr.run(36, learn=False)
        tr.flush()
    lifted = episodes = 0
    sim_eval = KinovaSim(1024, args.shape, auto_reset=False, horizon=30)
    qe, hqe = start_states(1024, args.shape, np.random.RandomState(args.seed + 1))
    qe, hqe = torch.as_tensor(qe), torch.as_tensor(hqe)
    t0 = time.perf_counter()
    t_eval = 0.0
    for it in range(0, args.steps, 60):
        if args.free_running:
            tr.run(60)                      # one persistent launch of 60 env-steps (+ 60 updates beside it): long launches keep the launch tail small
            tr.flush()
            c = tr.counts()
            d_ep, d_lift = c["episodes_finished"] - episodes, c["lifted"] - lifted
            episodes, lifted = c["episodes_finished"], c["lifted"]
        else:
            d_ep = d_lift = 0
            for _ in range(60):
                reward, done = tr.step()
                d_lift += int(((reward > 0) & done).sum())
                d_ep += int(done.sum())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0 - t_eval
        ls = tr.native.losses.tolist()
        ev = ""
        if args.eval_every and (it + 60) % args.eval_every == 0:
            te = time.perf_counter()
            tr.flush(finish_update=True)                     # the actor the next rollout step would use
            torch.cuda.synchronize()
            res = eval_policy(sim_eval, policy, sim_eval.reset(qe, hqe))
            ev = f"  eval (no noise, 1024 starts): lift success {res['num_success'] / 1024:.3f}"
            t_eval += time.perf_counter() - te
        print(f"step {it + 60:5d}  episodes {d_ep:7d}  training lift rate {d_lift / max(1, d_ep):.3f}  critic loss {ls[0]:9.3f}  {n * (it + 60) / dt:9.0f} env-steps/s{ev}")
    if args.dump_replay:
        tr.flush(finish_update=True)
        torch.cuda.synchronize()
        out = {}
        for name, rep, keep in (("agent", agent, slice(-1500, None)), ("expert", expert, slice(0, 400))):
            eps = rep.host_episodes()[keep] if rep is not None else []
            out[f"{name}_lens"] = np.array([len(e["reward"]) for e in eps])
            for f in ("state", "action", "next_state", "reward"):
                out[f"{name}_{f}"] = np.concatenate([np.asarray(e[f], dtype=np.float32) for e in eps]) if eps else np.zeros(0, np.float32)
        for name, net in (("actor", policy.actor), ("critic", policy.critic), ("actor_target", policy.actor_target), ("critic_target", policy.critic_target)):
            for k, v in net.state_dict().items():
                out[f"{name}.{k}"] = v.detach().cpu().numpy()
        out["updates"] = np.array(tr.updates)
        np.savez_compressed(args.dump_replay, **out)
        print("dumped", args.dump_replay, {k: v.shape for k, v in out.items() if k.endswith("_lens")}, "after", tr.updates, "updates")
    if args.save:
        tr.flush(finish_update=True)
        policy.save(args.save)
        print("saved", args.save + "_{actor,critic,actor_optimizer,critic_optimizer}")
    sim.close()
    sim_eval.close()


if __name__ == "__main__":
    main()
