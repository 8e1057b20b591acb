#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched Kinova gripper simulator + DDPG rollout on N MI355X (one process
per GPU).

One "step" = one env.step() (15 mj_step substeps + 82-d observation + reward/done) for every env of
the rank.  4096 envs per GPU (BASELINE metric), CubeS, 'normal' hand pose, env i starts at row
2 + (i mod 4498) of the no_noise start table, 30-step episodes with auto-reset.
  --mode ddpg (default, BASELINE config 3; config 4 with --gpus 8): actions a = clip(pi(s) + N(0, 0.08), 0, 0.8)
      from the 256-256 actor, scripted lift after check_grasp, transitions into the device replay, and ONE
      DDPGfD update per env-step on 64 episodes x 25 five-step windows (1600 rows); with N > 1 GPUs the
      gradients are averaged by RCCL all-reduce.  Nothing is skipped inside the timed region.
  --mode sim (BASELINE config 2 at the metric's env count): replays Generator(PCG64(1000 + i)).uniform(-0.8, 0.8)
      action streams resident in HBM; sim kernels only.

Timed region (round 3): in ddpg mode the policy is first trained for `--pretrain-updates` (1500) untimed env-steps - the hands
then close into contact-rich grasps and the envs' episode clocks are spread over the 30 phases - then W warm-up steps, then
`--repeats` (3) timed windows of EXACTLY K steps each, every one between a barrier + synchronize; the line is the window of median duration
(`timed_windows_ms_per_step` lists all).  (Rounds 1-2 timed the first episodes of a random policy, the cheapest regime.)

Rollout form (--rollout, default auto): the free-running kernel k_rollout (actor + 15 substeps + rays + observation + replay write in one
persistent launch of --chunk env-steps in which every wave loops over its own four envs, learner graphs beside it) when the widths are LDS-free (256-256 / 128-128 / 64-64) and
every workgroup is resident in one round; otherwise - config 5, 400-300, --eager, --serial-learner, or --rollout lockstep - one k_env_step
launch per env-step from HIP graphs.  `config.launch` says which one ran.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel k_rollout / k_env_step, HIP-event timed on the
launch stream inside this process), `mfma` (the learner's MLP kernels against the fp32 MFMA peak), `steady_state`
(a longer window of whole episodes right after the timed one: `value` should agree with it) and `cpu_baseline` (the fp64 CPU
oracle on the host cores, N=1 only).

Multi-GPU: `python bench.py --gpus N` starts N ranks itself (fresh child processes, spawned BEFORE this process touches
torch or HIP; rendezvous on 127.0.0.1) - the same thing the driver does with `python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N`; under a launcher (WORLD_SIZE set) --gpus must equal WORLD_SIZE.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

ALGO_BYTES_PER_ENV_STEP = 768          # read 236 B + write 532 B per env-step
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def shape_states(n: int, shape: str):
    """start states of the single-object workloads: BASELINE config 2's rows of the shape's no_noise table ('normal' class), through the
    reference reset's 5 cm correction (scenarios.reset_body_position: it moves the `object` geom's centre of a multi-geom object onto
    the row; a no-op for the README shapes) - or, where the reference ships no 'normal' table (the bowls), its empty-file rule."""
    import numpy as np
    from kinovagrasping_amd import scenarios
    if shape in scenarios.SHAPES:
        return scenarios.config2_states(n, shape)
    q = np.zeros((16, n))
    q[12] = 1.0
    rng = np.random.RandomState(2)
    tab = scenarios.start_coord_table(shape) if scenarios.has_start_table(shape, "normal") else None
    for i in range(n):
        cmd = tab[i % (len(tab) - 1)] if tab is not None else scenarios.fallback_start(shape, "normal", rng)
        q[9:12, i] = scenarios.reset_body_position(shape, cmd)
    return q, np.repeat(scenarios.hand_quat_for("normal")[:, None], n, axis=1)


def cpu_baseline(n_cores: int, budget_s: float = 12.0, shape: str = "CubeS"):
    """fp64 oracle ("port"), one env per thread, same workload (config-2 start rows / action streams)."""
    import numpy as np
    from kinovagrasping_amd import scenarios
    from kinovagrasping_amd.sim import SOLVER_ITERATIONS
    from oracle import ko_py as ko
    model = ko.OracleModel(scenarios.model_blob(shape))
    q0, hq = shape_states(n_cores, shape)
    acts = scenarios.config_actions(n_cores, 30)

    def worker(i):
        sim = ko.OracleSim(model, hq[:, i], solver_iterations=SOLVER_ITERATIONS)
        sim.env_reset(q0[:, i])
        steps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            for t in range(30):
                sim.env_step(acts[t][:, i])
            sim.env_reset(q0[:, i])
            steps += 30
        return steps, time.perf_counter() - t0

    with ThreadPoolExecutor(n_cores) as ex:
        res = list(ex.map(worker, range(n_cores)))
    total = sum(s / dt for s, dt in res)
    return {"value": round(total, 2), "unit": "env-steps/s", "cores": n_cores, "kind": "port",
            "sample": f"{n_cores} threads x ~{budget_s:.0f} s of 30-step {shape} episodes (config-2 rows/actions), fp64 oracle, "
                      f"{sum(s for s, _ in res)} env-steps total"}


def cpu_baseline_mujoco(budget_s: float = 8.0):
    """Only where the third-party `mujoco` package is importable (not on this image / the GPU boxes): real MuJoCo stepping
    the same CubeS model (rebuilt from the compiled blob, kinovagrasping_amd/mjcf_export.py) with config-2 actions, ONE
    thread; env-steps/s = mj_step rate / 15.  Returns None when mujoco is missing."""
    try:
        import mujoco
    except Exception:
        return None
    import numpy as np
    from kinovagrasping_amd import mjcf_export, scenarios
    q0, hq = scenarios.config2_states(1)
    m = mujoco.MjModel.from_xml_string(mjcf_export.to_mjcf(scenarios.model_blob("CubeS"), hq[:, 0]))
    d = mujoco.MjData(m)
    acts = scenarios.config_actions(1, 30)[:, :, 0]
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        d.qpos[:] = q0[:, 0]; d.qvel[:] = 0
        mujoco.mj_forward(m, d)
        for t in range(30):
            d.ctrl[:] = [0, 0.2932 * 0, 0, 0, acts[t][0], 0.2932, acts[t][1], acts[t][2], acts[t][3]]     # normal pose: wrist = slide_z
            for _ in range(15):
                mujoco.mj_step(m, d)
        steps += 30
    dt = time.perf_counter() - t0
    return {"value": round(steps / dt, 2), "unit": "env-steps/s", "cores": 1, "kind": "third-party mujoco " + mujoco.__version__,
            "sample": f"{steps} env-steps of 30-step CubeS episodes, one thread, model rebuilt from the compiled blob"}


def launch_ranks(n_ranks: int, argv) -> int:
    """Start `n_ranks` copies of this script as child processes (one per GPU, env-style rendezvous on 127.0.0.1) and wait.
    Nothing in this process has touched torch / HIP at this point, and no process that has is ever re-exec'ed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + list(argv), env=env))
    rc = 0
    try:
        # poll ALL children: a rank that dies while rank 0 is still blocked in the rendezvous or a collective must bring the
        # others down now, not after their NCCL timeout
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                rc = max(rc, abs(code))
            if rc and live:
                for q in live:
                    q.terminate()
                deadline = time.monotonic() + 10.0
                for q in live:
                    try:
                        q.wait(timeout=max(0.1, deadline - time.monotonic()))
                    except subprocess.TimeoutExpired:
                        q.kill()
                live = []
            elif live:
                time.sleep(0.05)
    except KeyboardInterrupt:
        for q in procs:
            if q.poll() is None:
                q.terminate()
        rc = 130
    return rc


MFMA_F32_PEAK_TFLOPS = 157.3           # MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_16x16x4_f32)


def update_flops(hidden, rows, n=5, state=82, action=4):
    """FLOPs of one DDPGfD update on `rows` sampled windows: MACs per sample A (actor) / C (critic), forward
    equivalents 2 (A + C) targets on 2R rows... = R (17 A + 20 C) MACs, x2 flops."""
    h1, h2 = hidden
    A = state * h1 + h1 * h2 + h2 * action
    C = (state + action) * h1 + h1 * h2 + h2
    return 2.0 * rows * (17 * A + 20 * C)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--mode", choices=["ddpg", "sim"], default="ddpg")
    ap.add_argument("--config", type=int, choices=[2, 3, 4, 5], default=None,
                    help="BASELINE config: 2 = --mode sim, 3 / 4 = DDPG training (4: with --gpus 8), 5 = DDPG training on the domain-"
                         "randomised set: 14 shapes x 3 hand poses, per-env mass / friction, 8192 envs per GPU")
    ap.add_argument("--shape", default="CubeS", help="object of the single-object workloads (configs 2 - 4; default CubeS = the metric's).  Any asset name: "
                    "the 14 README shapes, the primitives, or a multi-geom object (BottleS ... RBowlB: runs on libkinova_sim_mg.so)")
    ap.add_argument("--hidden", type=int, nargs=2, default=[256, 256])
    ap.add_argument("--serial-learner", action="store_true", help="run the learner update after the sim step instead of beside it")
    ap.add_argument("--eager", action="store_true", help="launch the rollout / learner ops one by one instead of replaying HIP graphs")
    ap.add_argument("--cohort", type=int, default=1, help="--config 5: draw the object once per this many consecutive envs (1 = per env, BASELINE's definition and "
                    "the headline; 16 = every shape's env count a multiple of the stepping kernel's group size: a labelled variant, not comparable)")
    ap.add_argument("--envs-per-gpu", type=int, default=None, help="default 4096 (the metric's env count); 8192 for --config 5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--init-policy", default="auto",
                    help="ddpg mode: start from a committed pre-trained 256-256 policy (the reference's 4-file checkpoint format, DDPGfD.py:371-382; default: "
                         "kinovagrasping_amd/assets/bench_policy/ddpg_256_256_* when present and the widths match) so that the untimed phase and the timed window start "
                         "from the SAME contact-rich regime in every run and every round - the throughput of this workload follows what the policy does "
                         "(`timed_window.regime`); 'none': random initialisation as in rounds 1-3")
    ap.add_argument("--pretrain-updates", type=int, default=None,
                    help="ddpg mode: untimed env-steps with learner updates BEFORE the warm-up, so that the timed steps see the trained policy's "
                         "contact-rich grasps and de-synchronised episode clocks instead of the cheap first episodes of a random policy (0 = time the "
                         "first episodes, as rounds 1-2 did); default 1500, or 300 when a pre-trained policy is loaded")
    ap.add_argument("--steady-updates", type=int, default=0, help="further learner updates before the steady_state window (whole episodes)")
    ap.add_argument("--rollout", choices=["auto", "lockstep", "free"], default="auto",
                    help="ddpg mode: 'free' = the free-running rollout kernel (ks_rollout / pipeline.AsyncTrainer, round 3): every stepping workgroup "
                         "loops over its 16 envs - in-kernel actor, 15 substeps, rays, observation, replay write - without waiting for other workgroups, "
                         "learner beside it (+6-18 %% at 4096 envs; training is not bit-reproducible run to run); 'lockstep' = one stepping launch per "
                         "env-step for all envs (pipeline.GraphedTrainer, rounds 1-2; bit-reproducible).  'auto' (default): free where it pays - the "
                         "learner is LDS-free (256-256) and the envs fit the GPU's CUs in one round of workgroups (<= 16 envs x CUs: config 3 / 4) - "
                         "else lockstep (config 5's 8192 envs, 400-300, --eager, --serial-learner).  The JSON line says which (`config.launch`).")
    ap.add_argument("--budget-ms", type=float, default=0.0, help="extra measurement (not the metric): launches with a TIME budget of this many ms instead of a step count "
                    "(free-running wave form only): what the exact count's launch tail costs")
    ap.add_argument("--repeats", type=int, default=3, help="timed windows of --steps env-steps each; the line reports the median window (all are listed)")
    ap.add_argument("--chunk", type=int, default=60, help="free-running rollout: at most this many env-steps per launch (learner and rollout streams meet between launches); "
                    "a launch lasts as long as its slowest workgroup, whose lead over the mean workgroup shrinks with the square root of the steps per launch; the "
                    "learner's stream is paced on the envs' step counters (pipeline.AsyncTrainer.run, kr_wait_min), so finished episodes are collected all the way whatever the length; 60 is kept as the default because the launch pattern of the pre-training decides which policy emerges "
                    "(launches of 100: ~10 % of the episodes end in a lift and the step is 8 % cheaper; launches of 60 + 40: 37 %, the regime all of this round's numbers are in)")
    ap.add_argument("--expert-prob", type=float, default=0.0,
                    help="ddpg mode: DDPGfD's demonstration mix (DDPGfD.py:232-254) - an expert ring is filled with one scripted 'combined'-controller "
                         "episode per env before training and every update samples int(64 (1 - p)) agent + the rest expert episodes (reference: 0.3)")
    ap.add_argument("--solver-iterations", type=int, default=None, help="Newton cap per substep (default: kinovagrasping_amd.sim.SOLVER_ITERATIONS)")
    ap.add_argument("--steady-steps", type=int, default=300, help="length of the steady_state window in env-steps (a multiple of the 30-step episode)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: start it as `python bench.py --gpus N` "
                 "or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")

    if args.config == 2:
        args.mode = "sim"
    if args.mode == "sim" and args.pretrain_updates is None:
        args.pretrain_updates = 0
    mixed = args.config == 5
    if args.envs_per_gpu is None:
        args.envs_per_gpu = 8192 if mixed else 4096

    import numpy as np
    import torch
    from kinovagrasping_amd import scenarios
    from kinovagrasping_amd.sim import SOLVER_ITERATIONS, KinovaSim
    iters = args.solver_iterations or SOLVER_ITERATIONS

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(1, int(os.environ.get("KS_VISIBLE_GPUS", "1000000")))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        # backend nccl = RCCL over xGMI; KS_DIST_BACKEND=gloo lets two ranks share one GPU (plumbing check only)
        backend = os.environ.get("KS_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    n = args.envs_per_gpu
    dev = torch.device("cuda", local_rank)
    # envs shard by global index: rank r owns envs [r*n, (r+1)*n); no data-path collective in the sim
    sl = slice(rank * n, (rank + 1) * n)
    if mixed:
        # config 5: 14 shapes x {normal, rotated, top} x mass / friction per env, all in ONE context and one stepping launch
        # (object drawn per env - BASELINE config 5 / SURVEY 8d; --cohort 16 is the labelled variant of round 4, see config5_states)
        oid_all, pose_all, q0_all, hq_all, mf_all = scenarios.config5_states(n * world, seed=5, cohort=args.cohort)
        sim = KinovaSim(n, scenarios.SHAPES, device=local_rank, auto_reset=True, horizon=30, solver_iterations=iters)
        reset_all = lambda: sim.reset(torch.as_tensor(q0_all[:, sl]), torch.as_tensor(hq_all[:, sl]), object_id=oid_all[sl], mass_friction=mf_all[:, sl])
    else:
        q0_all, hq_all = shape_states(n * world, args.shape)
        sim = KinovaSim(n, args.shape, device=local_rank, auto_reset=True, horizon=30, solver_iterations=iters)
        reset_all = lambda: sim.reset(torch.as_tensor(q0_all[:, sl]), torch.as_tensor(hq_all[:, sl]))
    obs0 = reset_all()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    updates = 0
    trainer = None
    learner_form = "none" if args.mode == "sim" else "autograd"
    if args.mode == "sim":
        base = scenarios.config_actions(min(n, 256), 30, base_seed=1000 + rank * n)
        acts = torch.as_tensor(np.tile(base, (1, 1, (n + base.shape[2] - 1) // base.shape[2]))[:, :, :n]).to(dev)
        step_fn = lambda t: sim.step(acts[t % 30])
    else:
        from kinovagrasping_amd.ddpgfd import DDPGfD
        from kinovagrasping_amd.replay import DeviceEpisodeReplay
        from kinovagrasping_amd.rollout import RolloutEngine
        torch.manual_seed(2)                                   # reference default seed (main_DDPGfD.py:884): identical replicas
        policy = DDPGfD(82, 4, 0.8, 5, batch_size=64, hidden=tuple(args.hidden), device=dev, capturable=not args.eager)
        init_policy = None
        if args.init_policy != "none":
            prefix = ROOT / "kinovagrasping_amd" / "assets" / "bench_policy" / "ddpg_256_256" if args.init_policy == "auto" else Path(args.init_policy)
            if Path(str(prefix) + "_actor").exists() and (args.init_policy != "auto" or tuple(args.hidden) == (256, 256)):
                policy.load(str(prefix), sync_targets=True)       # identical on every rank
                init_policy = str(prefix.relative_to(ROOT)) if prefix.is_relative_to(ROOT) else str(prefix)
            elif args.init_policy != "auto":
                sys.exit(f"bench.py: --init-policy {args.init_policy}: no such checkpoint ({prefix}_actor)")
        if args.pretrain_updates is None:
            args.pretrain_updates = 300 if init_policy else 1500
        torch.manual_seed(2 + rank)                            # exploration noise / window sampling differ per rank
        replay = DeviceEpisodeReplay(n, capacity=max(4 * n, 1024), horizon=30, device=dev)
        expert, expert_info = None, None
        if args.expert_prob > 0:
            # DDPGfD proper: demonstrations first (expert_data.py:690-921, one 'combined'-controller episode per env), kept in their own ring
            from kinovagrasping_amd.demonstrators import run_controller_episodes
            expert = DeviceEpisodeReplay(n, capacity=n, horizon=30, device=dev)
            demo = run_controller_episodes(sim, obs0.clone(), expert, horizon=30, mode="combined")
            expert_info = {"prob": args.expert_prob, "episodes": expert.count, "demonstrator": "combined", "demonstration_success_rate": round(float(demo["success"].float().mean()), 4),
                           "agent_episodes_per_batch": int(64 * (1 - args.expert_prob)), "expert_episodes_per_batch": 64 - int(64 * (1 - args.expert_prob))}
            obs0 = reset_all()
        if not args.eager:
            # rollout ops and the DDPGfD update replayed as HIP graphs; the simulator is launched between them and
            # the learner graph runs on a second stream beside the simulator kernel (kinovagrasping_amd/pipeline.py)
            from kinovagrasping_amd.pipeline import AsyncTrainer, GraphedTrainer
            eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
            eng.start(obs0)
            # how the library itself would schedule this context's env groups in a free-running launch (ks_rollout_plan; ADVICE r5: not re-derived here):
            # a fixed deal ("runs" / "round-robin": KS_ROLLOUT_DEAL, the multi-geom library's default) paces the launch at the workgroup with the most
            # groups unless they divide evenly
            plan, groups, wgs = sim.rollout_plan()
            fits = plan in ("waves", "workgroups", "queue") or groups % wgs == 0
            free_running = not args.serial_learner and (args.rollout == "free" or (args.rollout == "auto" and fits and tuple(args.hidden) in ((256, 256), (128, 128), (64, 64))))
            if free_running:
                try:
                    trainer = AsyncTrainer(sim, policy, replay, eng, batch_episodes=64, expert_replay=expert, expert_prob=args.expert_prob)
                except ValueError as e:                       # e.g. widths whose learner needs LDS: the lock-step trainer handles those
                    print(f"bench.py: --rollout free not available ({e}); using the lock-step trainer", file=sys.stderr)
                    free_running = False
            if not free_running:
                trainer = GraphedTrainer(sim, policy, replay, eng, batch_episodes=64, overlap=not args.serial_learner, expert_replay=expert,
                                         expert_prob=args.expert_prob)
            trainer.capture()
            learner_form = "lds-free fp32-mfma kernels" if trainer.native.lds_free else "library gemms + kr_* glue"

            def step_fn(t):
                nonlocal updates
                trainer.step()
                updates = trainer.updates
        else:
            gen = torch.Generator(device=dev).manual_seed(2 + rank)
            eng = RolloutEngine(sim, policy, replay, expl_noise=0.1, generator=gen)
            eng.start(obs0)
            main, side = torch.cuda.current_stream(dev), torch.cuda.Stream(dev)
            lgen = torch.Generator(device=dev).manual_seed(1002 + rank)
            acted = torch.cuda.Event()
            steps_done = 0

            def learner_update():
                nonlocal updates
                if steps_done < 31:                    # every env has finished an episode by step 30
                    return
                side.wait_event(acted)                 # weights are free once this step's actor forward is done
                with torch.cuda.stream(side), torch.enable_grad():
                    st, ac, ns, rw, nd, w = (replay.sample_mixed(expert, 64, args.expert_prob, generator=lgen) if expert is not None
                                             else replay.sample_batch_nstep(64, generator=lgen))
                    policy.train_on_batch(st, ac, ns, rw, w)
                updates += 1

            def step_fn(t):
                nonlocal steps_done
                if args.serial_learner:
                    eng.step()
                    learner_update()
                    main.wait_stream(side)
                else:
                    eng.step(after_act=lambda: acted.record(main), after_launch=learner_update, before_store=lambda: main.wait_stream(side))
                steps_done += 1

    # ddpg mode: the learner only has data once every env has finished an episode (30 steps) - if the requested warm-up is
    # shorter, prime the replay first so that EVERY timed step carries a learner update (nothing skipped in the timed region)
    priming = max(0, 36 - args.warmup) if args.mode == "ddpg" else 0
    free_running = trainer is not None and type(trainer).__name__ == "AsyncTrainer"
    k = 0
    rollout_ms = []                     # free-running: (launch duration by HIP events on the launch stream, env-steps) per launch
    all_launches = []                   # free-running: env-steps of EVERY k_rollout launch of this process, in order (tools/rollout_trace_join.py)

    def advance(n_steps, learn=True, timed=False):
        """n_steps env-steps of every env (+ as many learner updates)"""
        nonlocal k, updates
        if not free_running:
            for _ in range(n_steps):
                step_fn(k); k += 1
            return
        left = n_steps
        # (nccl without peer mapping: the per-update library all-reduce only executes between launches - short launches bound the learner's lag;
        #  pipeline.AsyncTrainer.library_allreduce_chunk)
        chunk = min(args.chunk, getattr(trainer, "library_allreduce_chunk", None) or args.chunk)
        while left > 0:
            c = min(left, chunk)
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                trainer.main.wait_stream(trainer.side)
                e0.record(trainer.main)
            trainer.run(c, learn=learn)
            all_launches.append(c)
            if timed:
                e1.record(trainer.main)
                rollout_ms.append((e0, e1, c))
            left -= c; k += c
        updates = trainer.updates

    if free_running:
        advance(priming, learn=False)
        trainer.flush()
    else:
        advance(priming)
    # untimed pre-training (round 3): the driver's `--steps 20 --warmup 5` used to time steps 6-25 of the FIRST episodes of a random
    # policy - fingers still closing, the cheapest third of an episode.  Now the policy is trained for `--pretrain-updates` updates
    # first: hands close into contact-rich grasps, episodes end at different steps (lift), so the envs' episode clocks are spread
    # over the 30 phases and ANY window of K steps is representative (`timed_window.episode_clock_histogram`).
    pretrained = 0
    if args.mode == "ddpg" and args.pretrain_updates > 0:
        while updates < args.pretrain_updates:
            advance(min(100, args.pretrain_updates - updates))
        pretrained = updates
    advance(args.warmup)
    barrier()
    clock_hist = torch.bincount(eng.t.clamp(0, 29), minlength=30).cpu().tolist() if args.mode == "ddpg" else None

    def ring_mark():
        """(committed episodes' ring head) - the regime of a window is read off the episodes that entered the replay ring during it"""
        return replay.head if args.mode == "ddpg" and not args.eager and trainer is not None else None

    def window_regime(head0, head1, count0=None, count1=None):
        """What the policy did in a window, so that rounds can be compared at equal regime (the throughput of this workload follows the
        policy: contact-rich grasps cost more): share of the episodes committed in the window that ended in a lift (their last stored
        reward carries the lift reward 50, main_DDPGfD.py:285-288), their mean stored length, and the contacts per env at its end."""
        if head0 is None:
            return None
        cap = replay.capacity
        k = (head1 - head0) % cap
        idx = (head0 + torch.arange(k, device=dev)) % cap
        L = replay.ep_len[idx].clamp(min=1)
        last = replay.ep_reward[idx, L - 1]
        ncon = sim.get_state()["ncon"].float()
        out = {"episodes_committed": int(k), "lift_fraction": round(float((last >= 50).float().mean()), 4) if k else None,
               "mean_stored_episode_length": round(float(L.float().mean()), 2) if k else None,
               "mean_contacts_per_env_at_end": round(float(ncon.mean()), 3), "max_contacts_per_env_at_end": int(ncon.max())}
        if count0 is not None:
            fin = count1["episodes_finished"] - count0["episodes_finished"]
            out["lift_fraction_all_finished_episodes"] = round((count1["lifted"] - count0["lifted"]) / max(1, fin), 4)
        return out

    # The timed region, `--repeats` times (the headline used to be ONE launch of 20 env-steps): each window is EXACTLY `--steps`
    # env-steps of every env (+ the learner's updates) between a barrier + synchronize on both sides; the line reports the window of MEDIAN duration
    # and lists all of them (`timed_windows_ms_per_step`).
    windows = []
    for rep in range(max(1, args.repeats)):
        head_t0 = ring_mark()
        cnt_t0 = trainer.counts() if free_running else None
        sim.kernel_time(reset=True)
        upd0 = updates
        t0 = time.perf_counter()
        advance(args.steps, timed=True)
        if trainer is not None:
            trainer.flush()                                    # the last step's deferred replay-ring update belongs to the timed work
        t_issue = time.perf_counter() - t0                 # host time to issue the timed steps (before the device catches up)
        barrier()
        dt = time.perf_counter() - t0
        kern_ms, launches = sim.kernel_time()
        regime_timed = window_regime(head_t0, ring_mark(), cnt_t0, trainer.counts() if free_running else None) if args.mode == "ddpg" else None
        if free_running:                     # per env-step: the persistent launches' durations / their env-steps
            tot = sum(e0.elapsed_time(e1) for e0, e1, _ in rollout_ms)
            kern_ms, launches = tot / max(1, sum(c for _, _, c in rollout_ms)), len(rollout_ms)
            timed_launches = "+".join(str(c) for _, _, c in rollout_ms)
            timed_first = len(all_launches) - len(rollout_ms)          # index of the first timed launch among all k_rollout launches
            rollout_ms.clear()
        timed_updates = updates - upd0
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = tt.item()
        windows.append(dict(dt=dt, t_issue=t_issue, kern_ms=kern_ms, launches=launches, regime_timed=regime_timed, timed_updates=timed_updates,
                            timed_launches=timed_launches if free_running else None, timed_first=timed_first if free_running else None))
    pick = sorted(windows, key=lambda w: w["dt"])[len(windows) // 2]
    dt, t_issue, kern_ms, launches, regime_timed, timed_updates = (pick[k_] for k_ in ("dt", "t_issue", "kern_ms", "launches", "regime_timed", "timed_updates"))
    if free_running:
        timed_launches, timed_first = pick["timed_launches"], pick["timed_first"]
    upd0 = updates - timed_updates
    # ---- the learner's MLP kernels against the fp32 MFMA peak: the update's graphs replayed ALONE (no simulator beside
    # them), HIP-graph launch gaps and the small glue kernels (sampling, loss gradient, Adam) included
    mfma = None
    if trainer is not None and updates > upd0:
        reps = 20
        barrier()
        t1 = time.perf_counter()
        with torch.cuda.stream(trainer.side):
            for _ in range(reps):
                trainer.g_head.replay()
                trainer._body()
            if getattr(trainer, "replica_sync", "") == "average-per-launch":
                trainer.average_replicas()                 # (these were local updates: the ranks meet again as they do at the end of a launch)
        trainer.main.wait_stream(trainer.side)
        barrier()
        upd_ms = (time.perf_counter() - t1) / reps * 1e3
        updates = trainer.updates
        rows = 64 * 25
        fl = update_flops(tuple(args.hidden), rows)
        tf = fl / (upd_ms * 1e-3) / 1e12
        mfma = {"bound": "mfma", "kernels": "learner update (k_mlp3_wave, k_mlp3_bwd_wave, k_wgrad_wave + glue) replayed alone",
                "flop_per_update": fl, "rows": rows, "update_ms_alone": round(upd_ms, 4), "achieved": round(tf, 3), "peak": MFMA_F32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 5), "dtype": "f32 (v_mfma_f32_16x16x4_f32, exact fp32)",
                "note": "in the pipeline the update runs on a second stream in the shadow of k_env_step (one wave per SIMD leaves the matrix "
                        "pipes idle) and is off the critical path; small-batch, latency-bound by design"}
    # ---- steady state: the same step after `--steady-updates` learner updates (the trained policy closes the hand into
    # contact-rich grasps, the stepping kernel grows with the contact count), window = a multiple of the 30-step episode
    steady = None
    if args.mode == "ddpg" and trainer is not None and args.steady_steps > 0:
        while updates < args.steady_updates:
            advance(min(100, args.steady_updates - updates))
        barrier()
        sim.kernel_time(reset=True)
        head_s0, cnt_s0 = ring_mark(), (trainer.counts() if free_running else None)
        u0, t2 = updates, time.perf_counter()
        advance(args.steady_steps, timed=True)
        trainer.flush()
        barrier()
        dts = time.perf_counter() - t2
        regime_steady = window_regime(head_s0, ring_mark(), cnt_s0, trainer.counts() if free_running else None)
        ks_ms, ks_n = sim.kernel_time()
        if free_running:
            tot = sum(e0.elapsed_time(e1) for e0, e1, _ in rollout_ms)
            ks_ms, ks_n = tot / max(1, sum(c for _, _, c in rollout_ms)), len(rollout_ms)
        if world > 1:
            tt = torch.tensor([dts], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dts = tt.item()
        steady = {"value": round(n * world * args.steady_steps / dts, 1), "unit": "env-steps/s", "steps": args.steady_steps,
                  "after_updates": u0, "learner_updates_timed": updates - u0, "ms_per_step": round(dts / args.steady_steps * 1e3, 4),
                  "k_env_step_avg_launch_ms": round(ks_ms, 4), "launches_timed": ks_n, "regime": regime_steady}
    # ---- opt-in extra (`--budget-ms`): the same loop with TIME-budgeted launches (ks_rollout_args.budget_ticks) - every wave steps its envs until a launch's
    # budget has passed, so nobody waits for the launch's slowest chain of env-steps; the rate is the env-steps actually done per second.  NOT the metric (the
    # envs no longer advance by the same count): it measures what the exact count costs - the launch tail.
    budgeted = None
    if args.budget_ms > 0 and free_running and world == 1 and getattr(trainer, "rollout_plan", "") == "waves":
        barrier()
        s0 = trainer.steps_total.clone()
        u0b, tb = trainer.updates, time.perf_counter()
        n_launch = 10
        per_launch = max(1, int(args.budget_ms / (dt / args.steps * 1e3)))         # updates beside a launch: about one per env-step of the mean env
        for _ in range(n_launch):
            trainer.run(4 * per_launch + 8, budget_ms=args.budget_ms, updates=per_launch)
        trainer.flush()
        barrier()
        dtb = time.perf_counter() - tb
        d = (trainer.steps_total - s0).float()
        budgeted = {"value": round(float(d.sum()) / dtb, 1), "unit": "env-steps/s", "launches": n_launch, "budget_ms_per_launch": args.budget_ms,
                    "env_steps_per_env": {"min": int(d.min()), "mean": round(float(d.mean()), 1), "max": int(d.max())}, "learner_updates": trainer.updates - u0b,
                    "note": "envs advance by time, not by count: not the metric; value / the line's value = what the launch tail of the exact count costs"}
        updates = trainer.updates
    status = sim.get_state()["status"]
    bad = int((status & 2).ne(0).sum().item())
    # replicas must hold bit-identical weights after the all-reduced updates: spread of two checksums over the ranks
    replica_spread = None
    if world > 1 and args.mode == "ddpg":
        if trainer is not None:
            replica_spread = trainer.replica_checksum_spread()
        else:
            chk = torch.stack([f(policy._flat_params[k].double()) for k in ("actor", "critic", "actor_target", "critic_target")
                               for f in (torch.sum, lambda x: x.abs().sum())])
            hi, lo = chk.clone(), chk.clone()
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            replica_spread = float((hi - lo).abs().max().item())
    if rank == 0:
        value = n * world * args.steps / dt
        achieved = ALGO_BYTES_PER_ENV_STEP * n / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (tools/pmc_run.sh);
        # the committed summary applies to the 4096-env workload only
        traffic, traffic_note, issue = None, "no PMC summary for this workload", None
        pmc = ROOT / "profiles" / ("r06_pmc_sim.json" if args.mode == "sim" else "r06_pmc_free.json" if free_running else "r06_pmc_ddpg.json")     # counters of THIS workload and THIS kernel (k_rollout's are per env-step)
        if pmc.exists() and n == 4096 and not mixed and args.shape == "CubeS":
            pj = json.loads(pmc.read_text())
            traffic, traffic_note = pj["hbm_bytes_per_launch"], pj["note"]
            pl = pj["per_launch"]
            if "SQ_WAVE_CYCLES" in pl:           # what actually bounds the kernel: one wave per SIMD, issue + LDS latency
                issue = {"bound": "valu-issue", "valu_busy_frac_of_wave_cycles": round(pl["SQ_ACTIVE_INST_VALU"] / pl["SQ_WAVE_CYCLES"], 3),
                         "waiting_frac_of_wave_cycles": round(pl["SQ_WAIT_ANY"] / pl["SQ_WAVE_CYCLES"], 3),
                         "valu_instructions_per_wave": round(pl["SQ_INSTS_VALU"] / pl["SQ_WAVES"]), "waves_per_simd": 1,
                         # useful lane-operations per launch against what 1024 SIMDs x 16 lanes could issue in the launch's cycles (GRBM_GUI_ACTIVE is summed
                         # over the 8 XCDs): the un-packed VALU issue rate actually used
                         "valu_issue_frac": (round(pl["SQ_INSTS_VALU"] * 64.0 * pj.get("valu_lane_efficiency", 0.0) / (1024 * 16 * pl["GRBM_GUI_ACTIVE"] / 8.0), 4)
                                             if pl.get("GRBM_GUI_ACTIVE") else None),
                         "valu_lane_efficiency": round(pj.get("valu_lane_efficiency", 0.0), 3),
                         "lds_bank_conflict_frac": round(pj.get("lds_bank_conflict_frac", 0.0), 3),
                         "l2_hit_rate": round(pj.get("l2_hit_rate", 0.0), 3),
                         "source": f"profiles/{pmc.name} (rocprofv3 --pmc; {pj['workload']})"}
        out = {
            "metric": "env-steps/sec (whole node) at 4096 envs/GPU", "value": round(value, 1), "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{n} envs/GPU DDPG training on the domain-randomised object set: 14 README shapes x {{normal, rotated, top}} hand "
                                    "poses (reference thresholds, no-noise tables, pose hand offsets), per-env mass U[0.05,0.15] kg and friction "
                                    "U[0.5,1.0] (" + ("everything drawn per env" if args.cohort == 1 else f"VARIANT: shape drawn per {args.cohort}-env cohort, everything else per env") + "), one simulator context / one stepping launch; 256-256 actor/critic, one DDPGfD update per env-step "
                                    "(BASELINE config 5; 65536 envs when n_gpus=8)") if mixed else
                                   (f"{n} envs/GPU DDPG training, 256-256 actor/critic, {args.shape} normal pose: actor inference + exploration noise + "
                                    "scripted lift in the loop, device replay, one DDPGfD update (64 episodes x 25 five-step windows) per "
                                    "env-step (BASELINE config 3; config 4 when n_gpus=8)") if args.mode == "ddpg" else
                                   (f"{n} envs/GPU {args.shape} normal-pose grasp sim, PCG64(1000+i) random-action rollout (BASELINE config 2 at the "
                                    "metric's env count); sim kernels only"),
                       "reset": ("the reference reset's 5 cm correction moves the `object` geom's centre onto the table row (scenarios.reset_body_position: a bottle starts "
                                  "buried and is lifted out by its floor contacts within an env-step); " if (not mixed and args.shape not in scenarios.SHAPES) else "") +
                                ("every env restarts from its own row of the reference's no_noise start table (obj_hand_coords/no_noise/train_coords), no orientation "
                                 "noise; hand slide offsets of the pose (hand_offsets='pose'): the workload SURVEY 8d defines, requested explicitly  [the env class's DEFAULTS are the "
                                 "reference's: KinovaGripperVecEnv.reset(with_noise=True) = its with_noise tables - which SURVEY N5 shows to be biased and swapped "
                                 "between classes - and hand_offsets='fresh-env' = the zero offsets its drivers end up with]"),
                       "mode": args.mode, "envs_per_gpu": n, "frame_skip": 15, "solver": f"newton, <= {iters} iterations per substep (early exit on convergence)", "hidden": list(args.hidden),
                       "learner_updates_timed": timed_updates if args.mode == "ddpg" else 0, "priming_steps": priming,
                       "launch": (("eager" if args.eager else ("free-running rollout kernel (ks_rollout), <= %d env-steps per launch (timed region: %s) + learner graphs" % (args.chunk, timed_launches)
                                                                if free_running else "hip-graphs, one stepping launch per env-step")) if args.mode == "ddpg" else "direct"),
                       "free_running": (dict(trainer.counts(), launch_steps=all_launches, first_timed_launch=timed_first, timed_launches=launches) if free_running else None),
                       "init_policy": (init_policy if args.mode == "ddpg" else None),
                       "learner": learner_form, "learner_stream_overlaps_rollout_stream": (getattr(trainer, "streams_overlap", None) if trainer is not None else None), "expert_mix": (expert_info if args.mode == "ddpg" else None),
                       "parallelism": f"env-shard x{world}" + ((" + replicas averaged per launch" if getattr(trainer, "replica_sync", "") == "average-per-launch" else " + grad all-reduce")
                                                                    if world > 1 and args.mode == "ddpg" else "")},
            "roofline": {"bound": "hbm", "kernel": "k_rollout (per env-step)" if free_running else "k_env_step", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 8), "traffic": traffic, "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * n,
                         "avg_launch_ms": round(kern_ms, 4), "launches_timed": launches,
                         "note": "algorithmic 768 B/env-step x envs per launch; the path is latency/VALU bound, not HBM bound (SURVEY 8d)",
                         "issue_bound": issue},
            "mfma": mfma,
            "steady_state": steady,
            "time_budgeted": budgeted,
            # every env runs the same 30-step episode clock (auto-reset), and an env-step costs more late in an episode (hands
            # closed, more contacts) than early: a window that is not whole episodes is not an average
            "timed_windows_ms_per_step": [round(w["dt"] / args.steps * 1e3, 4) for w in windows],      # every window of --steps env-steps; the line is the median one
            "timed_window": {"after_learner_updates": upd0, "pretrain_updates": pretrained, "steps": args.steps,
                             "episode_clock_histogram": clock_hist, "regime": regime_timed,
                             "note": ("timed after the untimed pre-training: trained policy, contact-rich grasps, env episode clocks spread over the 30 "
                                      "phases (histogram = envs per episode step at the start of the window) - the same regime as `steady_state`"
                                      if pretrained else
                                      "no pre-training (--pretrain-updates 0): all envs run the same episode clock and the first episodes of a random "
                                      "policy are the cheapest; quote `steady_state`") if args.mode == "ddpg" else "sim-only: 30-step random-action episodes, all envs in phase"},
            "nonfinite_envs": bad,
            "status_counts": {"contact_overflow": int((status & 1).ne(0).sum().item()), "nonfinite": bad,
                              "ray_pool_timeout": int((status & 4).ne(0).sum().item()),
                              "newton_ended_at_cap": int((status & 8).ne(0).sum().item())},
            "rccl": ({"ranks": world, "backend": os.environ.get("KS_DIST_BACKEND", "nccl"), "NCCL_ALGO": os.environ.get("NCCL_ALGO", "default"), "NCCL_PROTO": os.environ.get("NCCL_PROTO", "default"),
                      "allreduces_per_update": 2, "bytes_per_allreduce": int(policy._flat_params["critic"].numel() * 4),
                      "exchange": getattr(trainer, "exchange_note", None),
                      "replica_sync": getattr(trainer, "replica_sync", "per-update") if trainer is not None else None,   # "average-per-launch": KS_ASYNC_SYNC=average (opt-in local SGD)
                      "launch_chunk": (min(args.chunk, getattr(trainer, "library_allreduce_chunk", None) or args.chunk) if free_running else None),
                      "exchange_failed_call": (trainer.native.exchange.failed_epoch() if getattr(trainer, "native", None) is not None
                                               and trainer.native.exchange is not None else None)}
                     if world > 1 and args.mode == "ddpg" else None),
            "replica_weight_checksum_spread": replica_spread,
            "replica_checks_during_run": (getattr(trainer, "replica_checks", 0) if world > 1 else None),
            "host_issue_ms_per_step": round(t_issue / args.steps * 1e3, 4),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(os.cpu_count() or 1, shape="CubeS" if mixed else args.shape)
            mj = cpu_baseline_mujoco()                     # None unless the third-party mujoco package happens to be installed
            if mj is not None:
                out["cpu_baseline_mujoco"] = mj
        print(json.dumps(out))
    sim.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
