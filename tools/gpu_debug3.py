import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_gpu_parity import oracle_grasp_trajectory
from oracle import ko_py as ko
from kinovagrasping_amd.sim import KinovaSim
np.set_printoptions(precision=3, linewidth=220)
blob = __import__('kinovagrasping_amd.scenarios', fromlist=['x']).model_blob('CubeS')
cube = ko.OracleModel(blob)
hq, rec = oracle_grasp_trajectory(cube, n_sub=12)
def run(idx, prec, iters=6):
    n = len(idx)
    sim = KinovaSim(n, "CubeS", precision=prec, solver_iterations=iters)
    q0 = np.stack([rec[i][0][0] for i in idx], 1)
    sim.reset(torch.as_tensor(q0), torch.as_tensor(np.repeat(hq[:, None], n, 1)))
    sim.set_state(torch.as_tensor(q0), torch.as_tensor(np.stack([rec[i][0][1] for i in idx], 1)), torch.as_tensor(np.stack([rec[i][0][2] for i in idx], 1)))
    sim.substep(torch.as_tensor(np.stack([rec[i][1] for i in idx], 1)))
    st = sim.get_state(); torch.cuda.synchronize()
    qa = st['qacc_warmstart'].double().cpu().numpy()
    err = np.array([np.abs(qa[:, k] - rec[i][2][2]).max() for k, i in enumerate(idx)])
    sim.close()
    return err
print('fp64 uniform state0 x64', run([0]*64, 64)[:4])
print('fp64 uniform state8 x64', run([8]*64, 64)[:4])
print('fp64 mixed 0..11', run(list(range(12)), 64))
print('fp64 mixed 0..5', run(list(range(6)), 64))
print('fp64 single 0', run([0], 64))
print('fp64 mixed 0..11 iters 12', run(list(range(12)), 64, 12))
print('fp32 mixed 0..11', run(list(range(12)), 32))
