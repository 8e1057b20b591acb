#!/usr/bin/env python3
"""Experiment: does de-synchronising the batch pay?  G simulator contexts of N / G envs each, every one stepping on its own stream
with random actions (no cross-group dependency at all), against ONE context of N envs.  A launch lasts as long as its slowest wave
(1.6 - 1.9 x the median wave); with G groups a fast group does not wait for another group's stragglers.
usage (GPU box): GPU_MAX_HW_QUEUES=16 python tools/debug/group_streams.py [steps]"""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kinovagrasping_amd import scenarios  # noqa: E402
from kinovagrasping_amd.sim import KinovaSim  # noqa: E402

N = 4096
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
q0, hq = scenarios.config2_states(N)
base = scenarios.config_actions(256, 30, base_seed=1000)
acts_all = torch.as_tensor(np.tile(base, (1, 1, N // 256))).cuda()
# grasp-like regime: fingers closing half of the time
acts_all[:, 1:] = acts_all[:, 1:].abs()
for G in (1, 2, 4, 8, 16):
    n = N // G
    sims, streams, acts = [], [], []
    for g in range(G):
        s = KinovaSim(n, "CubeS", auto_reset=True, horizon=30)
        s.reset(torch.as_tensor(q0[:, g * n:(g + 1) * n]), torch.as_tensor(hq[:, g * n:(g + 1) * n]))
        sims.append(s); streams.append(torch.cuda.Stream()); acts.append(acts_all[:, :, g * n:(g + 1) * n].contiguous())
    torch.cuda.synchronize()

    def run(k):
        for t in range(k):
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    sims[g].step(acts[g][(t + 3 * g) % 30])       # groups start at different episode phases
    run(40)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"G={G:2d} ({n} envs per context): {N * steps / dt / 1e6:.3f} M env-steps/s, {dt / steps * 1e3:.4f} ms per round", flush=True)
    for s in sims:
        s.close()
