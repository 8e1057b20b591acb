#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched Kinova gripper simulator on N MI355X (one process per GPU).

One "step" = one env.step() (15 mj_step substeps + 82-d observation + reward/done) for every env of
the rank.  Workload: 4096 envs per GPU (BASELINE metric), CubeS, 'normal' hand pose, env i starts at
row 2 + (i mod 4498) of the no_noise start table and replays the action stream
Generator(PCG64(1000 + i)).uniform(-0.8, 0.8, (30, 4)) every 30-step episode (auto-reset) -- BASELINE
config 2 at the metric's env count.  Inputs (actions) are resident in HBM before the timed region.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel k_env_step, HIP-event timed on the
launch stream inside this process) and `cpu_baseline` (the fp64 CPU oracle on the host cores, N=1 only).
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


def cpu_baseline(n_cores: int, budget_s: float = 12.0):
    """fp64 oracle ("port"), one env per thread, same workload (config-2 start rows / action streams)."""
    import numpy as np
    from kinovagrasping_amd import scenarios
    from oracle import ko_py as ko
    model = ko.OracleModel((ROOT / "kinovagrasping_amd" / "assets" / "CubeS.ksm").read_bytes())
    q0, hq = scenarios.config2_states(n_cores)
    acts = scenarios.config_actions(n_cores, 30)

    def worker(i):
        sim = ko.OracleSim(model, hq[:, i], solver_iterations=6)
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
            "sample": f"{n_cores} threads x ~{budget_s:.0f} s of 30-step CubeS episodes (config-2 rows/actions), fp64 oracle, "
                      f"{sum(s for s, _ in res)} env-steps total"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    from kinovagrasping_amd import scenarios
    from kinovagrasping_amd.sim import KinovaSim

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n = args.envs_per_gpu
    # envs shard by global index: rank r owns envs [r*n, (r+1)*n); no data-path collective
    q0_all, hq_all = scenarios.config2_states(n * world)
    q0, hq = q0_all[:, rank * n:(rank + 1) * n], hq_all[:, rank * n:(rank + 1) * n]
    base = scenarios.config_actions(min(n, 256), 30, base_seed=1000 + rank * n)
    acts = torch.as_tensor(np.tile(base, (1, 1, (n + base.shape[2] - 1) // base.shape[2]))[:, :, :n]).cuda(local_rank)
    sim = KinovaSim(n, "CubeS", device=local_rank, auto_reset=True, horizon=30)
    sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for t in range(args.warmup):
        sim.step(acts[t % 30])
    barrier()
    sim.kernel_time(reset=True)
    t0 = time.perf_counter()
    for t in range(args.steps):
        sim.step(acts[(args.warmup + t) % 30])
    barrier()
    dt = time.perf_counter() - t0
    kern_ms, launches = sim.kernel_time()
    if world > 1:
        tt = torch.tensor([dt], device=f"cuda:{local_rank}", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    status = sim.get_state()["status"]
    bad = int((status & 2).ne(0).sum().item())
    if rank == 0:
        value = n * world * args.steps / dt
        achieved = ALGO_BYTES_PER_ENV_STEP * n / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        out = {
            "metric": "env-steps/sec (whole node) at 4096 envs/GPU", "value": round(value, 1), "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n} envs/GPU CubeS normal-pose grasp sim, random-action rollout (BASELINE config 2 at the metric's "
                                   "4096 envs/GPU), 15 substeps/env-step, 30-step episodes with auto-reset; sim kernels only "
                                   "(DDPG learner not in the loop yet)",
                       "envs_per_gpu": n, "frame_skip": 15, "solver": "newton x6", "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "kernel": "k_env_step", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 8), "traffic": None,
                         "avg_launch_ms": round(kern_ms, 4), "launches_timed": launches,
                         "note": "algorithmic 768 B/env-step x envs per launch; the path is latency/VALU bound, not HBM bound (SURVEY 8d)"},
            "nonfinite_envs": bad,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(os.cpu_count() or 1)
        print(json.dumps(out))
    sim.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
