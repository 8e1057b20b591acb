import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_gpu_parity import oracle_grasp_trajectory, run_teacher_forced
from oracle import ko_py as ko
from kinovagrasping_amd import scenarios
from kinovagrasping_amd.sim import KinovaSim
np.set_printoptions(precision=3, linewidth=200)
cube = ko.OracleModel(__import__('kinovagrasping_amd.scenarios', fromlist=['x']).model_blob('CubeS'))
hq, rec = oracle_grasp_trajectory(cube)
for prec in (64, 32):
    eq, ev, ncon, onc = run_teacher_forced(prec, cube, rec, hq)
    print('prec', prec, 'eq', eq)
    print('ncon mismatch idx', np.nonzero(ncon != onc)[0])
# smoke obs mismatch
n = 64
q0, hqs = scenarios.config2_states(n)
sim = KinovaSim(n, "CubeS", device=0)
obs0 = sim.reset(torch.as_tensor(q0), torch.as_tensor(hqs)).double().cpu().numpy().copy()
orc = [ko.OracleSim(cube, hqs[:, i], solver_iterations=6) for i in range(n)]
ref0 = np.stack([orc[i].env_reset(q0[:, i]) for i in range(n)])
bad = np.argwhere(np.abs(obs0 - ref0) > 2e-5 + 2e-4 * np.abs(ref0))
for e, j in bad[:40]:
    print('env', e, 'slot', j, 'gpu', obs0[e, j], 'ref', ref0[e, j])
