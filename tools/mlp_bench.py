"""Dev tool: kr_mlp3_forward against the torch modules (library GEMMs + elementwise kernels), eager launches timed with
HIP events.  usage: python tools/mlp_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from kinovagrasping_amd import mlp
from kinovagrasping_amd.ddpgfd import Actor, Critic

dev = torch.device("cuda", 0)


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for hidden in ((256, 256), (400, 300)):
    actor, critic = Actor(82, 4, 0.8, hidden).to(dev), Critic(82, 4, hidden).to(dev)
    la, lc = mlp.layers_of(actor), mlp.layers_of(critic)
    for n in (4096, 3200, 8000):
        s, a = torch.randn(n, 82, device=dev), torch.rand(n, 4, device=dev)
        out_a, out_q = torch.empty(n, 4, device=dev), torch.empty(n, 1, device=dev)
        with torch.no_grad():
            t_ta, t_tc = timed(lambda: actor(s)), timed(lambda: critic(s, a))
        t_fa = timed(lambda: mlp.mlp3_forward(la, s, act=mlp.ACT_SIGMOID, scale=0.8, out=out_a))
        t_fc = timed(lambda: mlp.mlp3_forward(lc, s, a, out=out_q))
        flop = 2.0 * n * (82 * hidden[0] + hidden[0] * hidden[1] + hidden[1] * 4)
        print(f"hidden {hidden} rows {n}: actor torch {t_ta:7.1f} us  fused {t_fa:6.1f} us ({flop / t_fa * 1e-6:5.1f} TFLOP/s);  "
              f"critic torch {t_tc:7.1f} us  fused {t_fc:6.1f} us")
