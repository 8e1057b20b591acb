"""Dev tool (diagnostic build -DKS_STAMP -DKS_STAMP_WG [-DKS_STAMP_HULL]): when does each workgroup of k_env_step run, and on which CU?"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from kinovagrasping_amd import scenarios
from kinovagrasping_amd.sim import KinovaSim
n = 4096
q0, hq = scenarios.config2_states(n)
base = scenarios.config_actions(256, 30)
acts = torch.as_tensor(np.tile(base, (1, 1, n // 256))).cuda()
sim = KinovaSim(n, "CubeS", auto_reset=True, horizon=30, contact_tap=True)
sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
closing = torch.tensor([0.0, 0.5, 0.5, 0.5], device="cuda").repeat(n, 1).t().contiguous()
MODE = sys.argv[1] if len(sys.argv) > 1 else "random"
eng = None
if MODE.startswith("policy"):
    # the bench's rollout (config 3): untrained actor + exploration noise + scripted lift; "policy:K" reports step K
    from kinovagrasping_amd.ddpgfd import DDPGfD
    from kinovagrasping_amd.rollout import RolloutEngine
    torch.manual_seed(2)
    policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=torch.device("cuda", 0))
    sim.close()
    sim = KinovaSim(n, "CubeS", auto_reset=True, horizon=30, contact_tap=True)
    obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
    LAST = int(MODE.split(":")[1]) if ":" in MODE else 45
    if MODE.startswith("policy-train"):
        # ... with the learner of the bench running (the actor changes as it is trained)
        from kinovagrasping_amd.replay import DeviceEpisodeReplay
        from kinovagrasping_amd.pipeline import GraphedTrainer
        policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=(256, 256), device=torch.device("cuda", 0), capturable=True)
        import os
        if os.environ.get("KS_INIT_POLICY"):            # e.g. kinovagrasping_amd/assets/bench_policy/ddpg_256_256: the bench's pre-trained regime
            policy.load(os.environ["KS_INIT_POLICY"], sync_targets=True)
        replay = DeviceEpisodeReplay(n, capacity=4 * n, horizon=30, device=torch.device("cuda", 0))
        eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
        eng.start(obs0)
        trainer = GraphedTrainer(sim, policy, replay, eng, batch_episodes=64, overlap=False)
        trainer.capture()
        eng.step = trainer.step
    else:
        eng = RolloutEngine(sim, policy, None, expl_noise=0.1)
        eng.start(obs0)
    MODE = "policy"
NSTEP = {"grasp": 22, "random": 12}.get(MODE, 0) or LAST + 1
for t in range(NSTEP):
    if t == NSTEP - 1:
        pre = {k: v.clone() for k, v in sim.get_state().items() if k in ("qpos", "qvel", "qacc_warmstart")}
    if eng is not None:
        eng.step()
    else:
        sim.step(closing if MODE == "grasp" else acts[t])
    st = sim.get_state(contacts=True)
    torch.cuda.synchronize()
    prof = st["contact"].reshape(-1, n)[:480].cpu().numpy().reshape(16, 30, n)[0]   # lane 0 of every env
    start, end, hw, xcc = prof[13], prof[14], prof[19].astype(np.int64), prof[20].astype(np.int64)
    wg = np.arange(n) // 16
    s0 = start.min()
    d = (end - start) % (1 << 22)
    st_rel = (start - s0) % (1 << 22)
    if (MODE == "random" and t >= 10) or (MODE == "grasp" and t >= 8 and (t % 3 == 0 or t == 21)) or (MODE == "policy" and (t % 5 == 0 or t == NSTEP - 1)):
        print(f"step {t}: per-env loop duration (100MHz ticks) min {d.min():.0f} mean {d.mean():.0f} max {d.max():.0f};  start spread max {st_rel.max():.0f};  last end {((end - s0) % (1<<22)).max():.0f}")
        cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
        key = xcc * 1000 + se * 100 + sh * 20 + cu
        uniq, cnt = np.unique(key[::16], return_counts=True)
        print("  distinct (xcc,se,sh,cu) used by the 256 workgroups:", len(uniq), " max WGs on one CU:", cnt.max())
        print(f"  kernel entry -> stepping loop (tables into LDS, state load), ticks: min {prof[23].min():.0f} mean {prof[23].mean():.0f} max {prof[23].max():.0f}")
        late = st_rel[::16] > 0.25 * d.mean()
        print("  workgroups starting late (> 25% of a loop):", int(late.sum()))
    if t >= NSTEP - 4:
        # persistence of an env's cost from one env-step to the next (would a cost-aware placement of envs in waves help?)
        fullh = st["contact"].reshape(-1, n)[:480].cpu().numpy().reshape(16, 30, n)
        hist = globals().setdefault("hist", [])
        hist.append((fullh[:, 21].max(0).copy(), fullh[:, 9].max(0).copy(), d.copy()))
        if len(hist) >= 3:
            a, b, c = hist[-3][0], hist[-2][0], hist[-1][0]
            top = lambda x: set(np.argsort(-x)[: n // 20])
            print(f"step {t}: #newton per env: corr(t-1, t) {np.corrcoef(b, c)[0, 1]:.2f}  corr(t-2, t) {np.corrcoef(a, c)[0, 1]:.2f};  top-5% envs shared with t-1: {len(top(b) & top(c)) / (n // 20):.2f}, with t-2: {len(top(a) & top(c)) / (n // 20):.2f}")
            # what a placement by the cost of step t-2 would do: waves = 4 envs; wave cost model = sum over its envs' excess iterations is not it -
            # the wave pays roughly max-per-substep; proxy: max of the four envs' #newton
            def wave_max(order):
                return c[order].reshape(-1, 4).max(1)
            ident = np.arange(n)
            rank = np.argsort(-a)                      # dealt round-robin: ranks w, w + n/4, ... share wave w
            dealt = rank.reshape(4, n // 4).T.reshape(-1)
            wm0, wm1 = wave_max(ident), wave_max(dealt)
            two = lambda order: int(((c[order].reshape(-1, 4) > 40).sum(1) >= 2).sum())
            print(f"   waves with >= 2 envs above 40 iterations: as placed {two(ident)}, dealt by the cost of t-2 {two(dealt)}")
    if t == NSTEP - 1:
        full = st["contact"].reshape(-1, n)[:480].cpu().numpy().reshape(16, 30, n)
        names = {0: "fk+dyn", 1: "collision", 2: "constraints", 3: "chol M", 5: "euler", 8: "c:plane", 9: "c:hull", 10: "c:merge", 13: "n:setup|wg-start", 14: "n:hessian|wg-end", 19: "n:update|hwid", 20: "post|xcc", 15: "n:chol", 16: "n:solve", 17: "n:pproj", 18: "n:ls", 21: "#newton", 22: "#ls"}
        order = np.argsort(-d)
        st2 = sim.get_state()
        nc = st2["ncon"].cpu().numpy()
        import os
        os.makedirs("gpurun_out", exist_ok=True)
        act = (eng.action_t if eng is not None else (closing if MODE == "grasp" else acts[NSTEP - 1])).cpu().numpy()
        np.savez("gpurun_out/slow_envs.npz", envs=order[:8], action=act[:, order[:8]], hand_quat=np.asarray(hq)[:, order[:8]], dur=d[order[:8]], **{k: v.cpu().numpy()[:, order[:8]] for k, v in pre.items()})
        wd = d[::4]                                  # one entry per wave (its four envs share the duration)
        print("  wave durations, percentiles 50/90/99/99.9/max:", [int(x) for x in np.percentile(wd, [50, 90, 99, 99.9, 100])],
              " waves above 1.2x / 1.4x the median:", int((wd > 1.2 * np.median(wd)).sum()), int((wd > 1.4 * np.median(wd)).sum()))
        seen = set()
        for e in order:
            if e // 4 in seen:
                continue
            seen.add(e // 4)
            envs4 = range(e // 4 * 4, e // 4 * 4 + 4)
            print(f"    wave {e // 4}: {d[e]:.0f} ticks; per env (gjk supports of the busiest lane, #newton, hull cycles):",
                  [(int(full[:, 25, x].max()), int(full[:, 21, x].max()), int(full[:, 9, x].max())) for x in envs4])
            if len(seen) >= 12:
                break
        for rank in (0, 1, 2, 3, n // 2, n - 1):
            e = order[rank]
            p = full[:, :, e].max(0)
            print(f"  env {e} (rank {rank}) duration {d[e]:.0f} ticks, ncon(last) {nc[e]}, total cyc {p[6]:.0f}: " + ", ".join(f"{names[k]} {p[k]:.0f}" for k in (0, 1, 8, 9, 10, 2, 3, 5, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22)) + f" | gjk calls {full[:, 24, e].sum():.0f} gjk supports {full[:, 25, e].sum():.0f} busiest lane: pre-GJK cycles {full[:, 26, e].max():.0f} gjk_distance cycles {full[:, 27, e].max():.0f} gjk sup {full[:, 25, e].max():.0f} gjk calls {full[:, 24, e].max():.0f}; support cycles {full[:, 28, e].max():.0f} closest cycles {full[:, 29, e].max():.0f}")
