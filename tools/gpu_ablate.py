"""Dev tool: time k_env_step with subsets of the contact pairs / solver iterations (ablation)."""
import sys, io, time, numpy as np, torch
sys.path.insert(0, '.')
from kinovagrasping_amd import model_compiler as mc, scenarios
from kinovagrasping_amd.sim import KinovaSim
import tempfile, pathlib
M = mc.read_blob(scenarios.model_blob('CubeS'))
n = 4096
q0, hq = scenarios.config2_states(n)
base = scenarios.config_actions(256, 30)
acts = torch.as_tensor(np.tile(base, (1, 1, n // 256))).cuda()
def run(npairs, iters, label, lpw=0):
    M2 = dict(M); M2['pairs'] = M['pairs'][:npairs].copy()
    blob = mc.blob_bytes(M2)
    sim = KinovaSim(n, blob, auto_reset=True, horizon=30, solver_iterations=iters, envs_per_wave=lpw)
    sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
    for t in range(5): sim.step(acts[t])
    torch.cuda.synchronize(); sim.kernel_time(reset=True)
    for t in range(15): sim.step(acts[5 + t])
    ms, k = sim.kernel_time()
    st = sim.get_state(); ncon = st['ncon'].float().mean().item()
    print(f"{label:40s} pairs {npairs:2d} iters {iters}  k_env_step {ms:8.3f} ms  mean ncon {ncon:.2f}", flush=True)
    sim.close()
run(30, 6, 'all pairs'); run(15, 6, 'no hand-hand'); run(8, 6, 'no hand-ground'); run(1, 6, 'object-ground only')
run(30, 2, 'all pairs, 2 newton'); run(1, 1, 'object-ground only, 1 newton'); run(30, 3, 'all pairs 3 newton')
