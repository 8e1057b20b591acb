import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_gpu_parity import oracle_grasp_trajectory
from tests.native_build import Lane
from oracle import ko_py as ko
from kinovagrasping_amd.sim import KinovaSim
np.set_printoptions(precision=3, linewidth=220)
blob = __import__('kinovagrasping_amd.scenarios', fromlist=['x']).model_blob('CubeS')
cube = ko.OracleModel(blob)
hq, rec = oracle_grasp_trajectory(cube, n_sub=12)
n = len(rec)
sim = KinovaSim(n, "CubeS", precision=64, solver_iterations=6)
q0 = np.stack([r[0][0] for r in rec], 1)
sim.reset(torch.as_tensor(q0), torch.as_tensor(np.repeat(hq[:, None], n, 1)))
sim.set_state(torch.as_tensor(q0), torch.as_tensor(np.stack([r[0][1] for r in rec], 1)), torch.as_tensor(np.stack([r[0][2] for r in rec], 1)))
sim.substep(torch.as_tensor(np.stack([r[1] for r in rec], 1)))
st = sim.get_state()
torch.cuda.synchronize()
lane = Lane(blob, 64)
for i in range(n):
    qa = st['qacc_warmstart'].cpu().numpy()[:, i]
    qv = st['qvel'].cpu().numpy()[:, i]
    print(i, 'qacc err vs oracle', np.abs(qa - rec[i][2][2]), 'ncon', rec[i][3])
    lp = lane.substep(*rec[i][0], rec[i][1], hq)
    print('   lane qacc err vs oracle', np.abs(lp[2] - rec[i][2][2]).max(), 'gpu qvel err', np.abs(qv - rec[i][2][1]).max())
