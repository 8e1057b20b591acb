"""Dev tool: how long does the actor forward take on an idle GPU (the action selection sits on the critical chain between two
stepping launches)?  Back-to-back launches, HIP events around the batch."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from kinovagrasping_amd import mlp
torch.manual_seed(0)
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = 256
layers = [(torch.randn(H, 82, device=dev) * 0.1, torch.zeros(H, device=dev)), (torch.randn(H, H, device=dev) * 0.05, torch.zeros(H, device=dev)),
          (torch.randn(4, H, device=dev) * 0.05, torch.zeros(4, device=dev))]
x = torch.randn(n, 82, device=dev)
out = torch.empty(n, 4, device=dev)

def timeit(label, fn, reps=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{label:40s} {e0.elapsed_time(e1) / reps * 1e3:7.1f} us per launch")

timeit("k_mlp3 (LDS, 4 waves / 16 rows)", lambda: mlp.mlp3_forward(layers, x, act=mlp.ACT_SIGMOID, scale=0.8, out=out))
for w in (4, 2, 0):
    os.environ["KS_MLP_SPLIT"] = str(w)
    timeit(f"LDS-free, split over {w} waves" if w else "LDS-free, one wave / 16 rows", lambda: mlp.mlp3_forward(layers, x, act=mlp.ACT_SIGMOID, scale=0.8, out=out, shadow=True))
