"""Per-env drift of the fp32 kernel from the fp64 oracle after 6 env steps of a closing grasp, for one shape
(the measurement behind tests/test_gpu_parity.py::test_all_fourteen_shapes_track_the_oracle).
usage: python tests/studies/shape_drift.py Vase1B [fp64]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kinovagrasping_amd import scenarios, sim as ks   # noqa: E402
from oracle import ko_py as ko                        # noqa: E402

sh = sys.argv[1] if len(sys.argv) > 1 else "Vase1B"
prec = 64 if "fp64" in sys.argv[2:] else 32
per = 8
tab = scenarios.start_coord_table(sh)
idx = np.linspace(0, len(tab) - 1, per).astype(int)
q0 = np.zeros((16, per)); q0[12] = 1; q0[9:12] = tab[idx].T
hq = np.repeat(scenarios.hand_quat_for("normal")[:, None], per, 1)
act = np.repeat(np.array([0.0, 0.6, 0.5, 0.7])[:, None], per, 1)
sim = ks.KinovaSim(per, sh, precision=prec)
sim.reset(torch.as_tensor(q0), torch.as_tensor(hq))
for t in range(6):
    sim.step(torch.as_tensor(act))
torch.cuda.synchronize()
qg = sim.get_state()["qpos"].double().cpu().numpy()
model = ko.OracleModel(scenarios.model_blob(sh))
rel = []
for i in range(per):
    o = ko.OracleSim(model, hq[:, i], solver_iterations=6)
    o.env_reset(q0[:, i])
    for t in range(6):
        o.env_step(act[:, i])
    qo = o.view("qpos")
    rel.append(np.abs(qg[:, i] - qo).max() / max(1e-3, np.abs(qo).max()))
print(sh, prec, np.array(rel))
