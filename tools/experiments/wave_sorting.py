#!/usr/bin/env python3
"""Dev tool (GPU box, -DKS_STAMP build via KS_LIB): what would grouping the 16 envs of a workgroup into its 4 waves BY WORKLOAD gain?  A wave runs as long as its
slowest env (Newton iterations, narrow-phase passes).  From the per-env Newton iteration counts and live-pair counts of consecutive env-steps in the bench's policy
regime: the wave maximum as it is (envs 4k .. 4k+3 of a workgroup share a wave), with the envs sorted by the SAME step's count (an upper bound) and sorted by the
PREVIOUS step's count (what a kernel could do).  Outcome: the mean wave would gain 16 %, but the workgroup waits for its SLOWEST wave (the four run side by side), and
that one hardly changes - see the last lines of the output; built and measured, not kept."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from pathlib import Path
from kinovagrasping_amd import scenarios
from kinovagrasping_amd.sim import KinovaSim
from kinovagrasping_amd.ddpgfd import DDPGfD
from kinovagrasping_amd.rollout import RolloutEngine
from kinovagrasping_amd.replay import DeviceEpisodeReplay
from kinovagrasping_amd.pipeline import GraphedTrainer
n = 4096
q0, hq = scenarios.config2_states(n)
torch.manual_seed(2)
sim = KinovaSim(n, "CubeS", auto_reset=True, horizon=30, contact_tap=True)
obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=torch.device("cuda", 0), capturable=True)
policy.load(str(Path("kinovagrasping_amd/assets/bench_policy/ddpg_256_256")), sync_targets=True)
replay = DeviceEpisodeReplay(n, capacity=4 * n, horizon=30, device=torch.device("cuda", 0))
eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
eng.start(obs0)
trainer = GraphedTrainer(sim, policy, replay, eng, batch_episodes=64, overlap=False)
trainer.capture()
for t in range(300):
    trainer.step()
hist = []
for t in range(300, 340):
    trainer.step()
    st = sim.get_state(contacts=True)
    torch.cuda.synchronize()
    prof = st["contact"].reshape(-1, n)[:480].cpu().numpy().reshape(16, 30, n)
    hist.append((prof[0, 21].copy(), prof[:, 24].sum(0), st["ncon"].cpu().numpy().astype(float), prof[0, 6].copy()))   # Newton iterations, live hull pairs (summed over substeps), ncon at the end, total cycles
def wave_max(x, order):
    y = np.take_along_axis(x.reshape(-1, 16), order, 1).reshape(-1, 4)
    return y.max(1).mean()
ident = np.tile(np.arange(16), (n // 16, 1))
for name, k in (("Newton iterations per env-step", 0), ("live hull pairs per env-step", 1)):
    asis = np.mean([wave_max(h[k], ident) for h in hist[1:]])
    best = np.mean([wave_max(h[k], np.argsort(h[k].reshape(-1, 16), 1)) for h in hist[1:]])
    prev = np.mean([wave_max(hist[i][k], np.argsort(hist[i - 1][k].reshape(-1, 16), 1)) for i in range(1, len(hist))])
    print(f"{name}: mean per env {np.mean([h[k].mean() for h in hist[1:]]):.2f}; wave maximum as it is {asis:.2f}; envs of a workgroup sorted by this step's count {best:.2f}; "
          f"sorted by the previous env-step's count {prev:.2f}")
# which cheap key predicts the next env-step's work?  (ncon at the end of the previous env-step is what the product build already stores per env)
for key_name, kk in (("previous ncon", 2), ("previous Newton iterations", 0), ("previous live hull pairs", 1), ("previous Newton iterations + live hull pairs", -1)):
    out = []
    for name, k in (("Newton", 0), ("hull pairs", 1)):
        vals = []
        for i in range(1, len(hist)):
            key = hist[i - 1][0] + hist[i - 1][1] if kk == -1 else hist[i - 1][kk]
            vals.append(wave_max(hist[i][k], np.argsort(key.reshape(-1, 16), 1, kind="stable")))
        out.append(f"{name} {np.mean(vals):.2f}")
    print(f"sorted by {key_name}: wave maximum " + ", ".join(out))
# one scalar key per env: which mix of the two counters minimises the modelled wave cost 17 k cycles x max(Newton iterations) + 9.6 k cycles x max(live hull pairs)?
def cost(order_of):
    tot = 0.0
    for i in range(1, len(hist)):
        o = order_of(i)
        tot += 17.0 * wave_max(hist[i][0], o) + 9.6 * wave_max(hist[i][1], o)
    return tot / (len(hist) - 1)
print(f"modelled wave cost per env-step (k cycles): as it is {cost(lambda i: ident):.0f}; sorted by this step's own total {cost(lambda i: np.argsort((17 * hist[i][0] + 9.6 * hist[i][1]).reshape(-1, 16), 1)):.0f}")
for wn, wh in ((1, 0), (0, 1), (17, 9.6), (17, 20), (17, 5), (1, 1)):
    print(f"   sorted by the previous env-step's {wn} x Newton + {wh} x hull: {cost(lambda i: np.argsort((wn * hist[i - 1][0] + wh * hist[i - 1][1]).reshape(-1, 16), 1, kind='stable')):.0f}")
# an exponential average of the key over the last env-steps instead of the last one alone
for beta in (0.5, 0.75):
    ema = np.zeros(n); tot = 0.0
    for i in range(len(hist)):
        if i >= 1:
            o = np.argsort(ema.reshape(-1, 16), 1, kind="stable")
            tot += 17.0 * wave_max(hist[i][0], o) + 9.6 * wave_max(hist[i][1], o)
        ema = beta * ema + (1 - beta) * (17 * hist[i][0] + 9.6 * hist[i][1])
    print(f"   sorted by an exponential average (beta {beta}) of 17 x Newton + 9.6 x hull: {tot / (len(hist) - 1):.0f}")
# BUT the four waves of a workgroup run SIDE BY SIDE (one per SIMD) and the workgroup's env-step ends with its slowest wave: what counts is the maximum over
# the workgroup's waves, not their mean.  Modelled workgroup cost = max over its 4 waves of (17 k x max Newton + 9.6 k x max hull); no arrangement can beat the
# workgroup's single slowest env (17 k x its Newton + 9.6 k x its hull pairs).
def wg_cost(order_of):
    tot = 0.0
    for i in range(1, len(hist)):
        o = order_of(i)
        a = np.take_along_axis(hist[i][0].reshape(-1, 16), o, 1).reshape(-1, 4, 4).max(2)
        h = np.take_along_axis(hist[i][1].reshape(-1, 16), o, 1).reshape(-1, 4, 4).max(2)
        tot += (17.0 * a + 9.6 * h).max(1).mean()
    return tot / (len(hist) - 1)
lower = np.mean([(17.0 * hist[i][0] + 9.6 * hist[i][1]).reshape(-1, 16).max(1).mean() for i in range(1, len(hist))])
print(f"modelled WORKGROUP cost per env-step (k cycles): as it is {wg_cost(lambda i: ident):.0f}; slots sorted by the previous env-step's hull count {wg_cost(lambda i: np.argsort(hist[i - 1][1].reshape(-1, 16), 1, kind='stable')):.0f}; "
      f"sorted by this step's own total {wg_cost(lambda i: np.argsort((17 * hist[i][0] + 9.6 * hist[i][1]).reshape(-1, 16), 1)):.0f}; lower bound (the slowest env alone) {lower:.0f}")
print("-> measured with the sorted dealing built into k_env_step / k_rollout (round 5, not kept): default training 3.22 -> 3.21 M env-steps/s, sim-only 4.15 -> 3.92 M.")
