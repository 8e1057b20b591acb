"""Round-5 study helper: the oracle's replay of Pose_file_2 (tests/old_env.replay_recording), cached under /tmp."""
import pickle, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
CACHE = Path("/tmp/replay_ms.pkl")


def load():
    from tests import old_env
    pf2 = np.load(ROOT / "tests/golden/mujoco_recorded.npz")["pose_file_2"]
    if CACHE.exists():
        rows, us, states = pickle.load(open(CACHE, "rb"))
    else:
        rows, us, states = old_env.replay_recording(pf2)
        pickle.dump((rows, us, states), open(CACHE, "wb"))
    return pf2, rows, us, states
