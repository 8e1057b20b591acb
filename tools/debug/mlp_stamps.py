"""Dev tool (diagnostic build -DKS_MLP_STAMP as KS_LIB): wall-clock stamps (100 MHz) of workgroup 0 through k_mlp3_pre."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from kinovagrasping_amd import mlp, sim as _sim
dev = torch.device("cuda", 0)
n, H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 256
layers = [(torch.randn(H, 82, device=dev) * 0.1, torch.zeros(H, device=dev)), (torch.randn(H, H, device=dev) * 0.05, torch.zeros(H, device=dev)),
          (torch.randn(4, H, device=dev) * 0.05, torch.zeros(4, device=dev))]
x, out = torch.randn(n, 82, device=dev), torch.empty(n, 4, device=dev)
lib = _sim.load_library()
buf = (ctypes.c_longlong * 16)()
for it in range(6):
    mlp.mlp3_forward(layers, x, act=mlp.ACT_SIGMOID, scale=0.8, out=out)
    torch.cuda.synchronize()
    lib.kr_debug_mlp_stamps(buf)
    t = list(buf)[:6]
    print("entry -> requests issued -> values in (layer 1 starts) -> layer 2 -> layer 3 -> end, ticks of 10 ns:", [t[i + 1] - t[i] for i in range(5)], "total", t[5] - t[0])
