#!/usr/bin/env python3
"""Round 5 (GPU box; VERDICT r4 next #4: "first record the throughput of k_env_step<double> as it is - nobody has"): stepping rate of the fp64
instantiation of the stepping kernel (the parity instrument: precision=64 contexts) against the fp32 product under the same protocol - 4096 CubeS
envs, config-2 start rows, random actions in [0, 0.8] (the hand closes: contacts), auto-reset, lock-step ks_step.
usage: python tools/r05/fp64_rate.py [n_envs] > profiles/r05_fp64_rate.txt"""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kinovagrasping_amd import scenarios
from kinovagrasping_amd.sim import KinovaSim

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for prec in (32, 64):
    sim = KinovaSim(n, "CubeS", auto_reset=True, horizon=30, precision=prec)
    q0, hq = scenarios.config2_states(n)
    sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
    g = torch.Generator(device="cuda").manual_seed(0)
    dt_ = torch.float32 if prec == 32 else torch.float64
    acts = [(torch.rand((4, n), device="cuda", generator=g) * 0.8).to(dt_) for _ in range(8)]
    for t in range(10):
        sim.step(acts[t % 8])
    torch.cuda.synchronize()
    t0 = time.time()
    K = 60 if prec == 32 else 30
    for t in range(K):
        sim.step(acts[t % 8])
    torch.cuda.synchronize()
    dt = time.time() - t0
    st = sim.get_state()
    print(f"precision {prec}: {n} envs, {K} env-steps: {n * K / dt / 1e6:.3f} M env-steps/s ({dt / K * 1e3:.3f} ms per env-step), status bits {int(st['status'].max())}, "
          f"mean contacts {float(st['ncon'].float().mean()):.2f}", flush=True)
    sim.close()
