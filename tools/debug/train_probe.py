#!/usr/bin/env python3
"""Why does the example's DDPGfD run stop lifting?  Lock-step trainer, 4096 envs, prints the mean action / lift statistics with and without the expert mix.
usage: python tools/debug/train_probe.py [expert_prob]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kinovagrasping_amd import scenarios  # noqa: E402
from kinovagrasping_amd.ddpgfd import DDPGfD  # noqa: E402
from kinovagrasping_amd.demonstrators import run_controller_episodes  # noqa: E402
from kinovagrasping_amd.pipeline import GraphedTrainer  # noqa: E402
from kinovagrasping_amd.replay import DeviceEpisodeReplay  # noqa: E402
from kinovagrasping_amd.rollout import RolloutEngine  # noqa: E402
from kinovagrasping_amd.sim import KinovaSim  # noqa: E402

prob = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
n = 4096
dev = torch.device("cuda", 0)
torch.manual_seed(2)
rng = np.random.RandomState(2)
q0, hq = scenarios.config2_states(n)
q0, hq = torch.as_tensor(q0), torch.as_tensor(hq)
expert = None
if prob > 0:
    expert = DeviceEpisodeReplay(n, capacity=4096, device=dev)
    sim = KinovaSim(n, "CubeS", auto_reset=False, horizon=30)
    out = run_controller_episodes(sim, sim.reset(q0, hq), expert, mode="combined")
    print("expert success", out["success"].float().mean().item(), "episodes", expert.count)
    L = expert.ep_len[: expert.count]
    A = expert.ep_action[: expert.count]
    R = expert.ep_reward[: expert.count]
    print("expert episode length mean", L.float().mean().item(), "min", L.min().item(), "max", L.max().item())
    m = (torch.arange(A.shape[1], device=dev)[None, :] < L[:, None])
    print("expert action mean per dim", (A * m[..., None]).sum((0, 1)) / m.sum(), " reward sum per episode mean", (R * m).sum(1).mean().item())
    sim.close()
sim = KinovaSim(n, "CubeS", auto_reset=True, horizon=30)
policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=dev, capturable=True)
agent = DeviceEpisodeReplay(n, capacity=4 * n, device=dev)
eng = RolloutEngine(sim, policy, agent)
eng.start(sim.reset(q0, hq))
tr = GraphedTrainer(sim, policy, agent, eng, batch_episodes=64, expert_replay=expert, expert_prob=prob if prob > 0 else 0.3)
tr.capture()
lift = ep = 0
for it in range(3000):
    reward, done = tr.step()
    lift += int(((reward > 0) & done).sum()); ep += int(done.sum())
    if (it + 1) % 150 == 0:
        tr.flush()
        with torch.no_grad():
            a = eng.action.float().mean(0).tolist()
            lifting = eng.lifting.float().mean().item()
            s = eng.obs[:512].float()
            pa = policy.actor(s)
            q = policy.critic(s, pa).mean().item()
        print(f"step {it + 1:5d} episodes {ep:6d} lift rate {lift / max(1, ep):.3f}  mean action {[round(x, 3) for x in a]}  envs in scripted lift {lifting:.2f}  Q(s, pi(s)) {q:8.2f}  losses {[round(x, 3) for x in tr.native.losses.tolist()]}")
        lift = ep = 0
tr.flush(finish_update=True)
torch.cuda.synchronize()
if expert is not None:
    with torch.no_grad():
        st, ac, nx, rw, nd, wt = agent.sample_mixed(expert, 64, prob)
        R = st.shape[0]
        b_agent = int(64 * (1 - prob)) * 25
        for name, sl in (("agent rows", slice(0, b_agent)), ("expert rows", slice(b_agent, R))):
            w = wt[sl] > 0
            s0, a0 = st[sl][w][:, 0], ac[sl][w][:, 0]
            q_data = policy.critic(s0, a0).squeeze(1)
            pa = policy.actor(s0)
            q_pi = policy.critic(s0, pa).squeeze(1)
            a_req = pa.clone().requires_grad_(True)
            with torch.enable_grad():
                policy.critic(s0, a_req).sum().backward()
            print(f"{name}: {int(w.sum())} live of {w.numel()}; reward sum over the 5-step windows mean {rw[sl][w].sum(1).mean():.3f}; data action mean {a0.mean(0).tolist()}")
            print(f"    Q(s, a_data) mean {q_data.mean():.3f}   Q(s, pi(s)) mean {q_pi.mean():.3f}   pi(s) mean {pa.mean(0).tolist()}   dQ/da at pi(s) mean {a_req.grad.mean(0).tolist()}")
        # Q along the wrist channel at expert states: what does the critic think of lifting the hand while the fingers close?
        s0 = st[b_agent:][wt[b_agent:] > 0][:, 0][:256]
        for wv in (0.0, 0.03, 0.06, 0.1, 0.2, 0.4):
            a = torch.tensor([wv, 0.5, 0.5, 0.5], device=dev).expand(s0.shape[0], 4)
            print(f"    expert states, a = [{wv}, .5, .5, .5]: Q mean {policy.critic(s0, a).mean():.3f}")
sim.close()
