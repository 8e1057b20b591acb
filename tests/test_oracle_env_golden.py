"""The oracle's env layer (oracle/ko_env.c) against golden vectors produced by the reference's own
Python (tests/golden/gen_golden_env.py -> tests/golden/env_layer.npz).  CPU only."""
import numpy as np
import pytest

from oracle import ko_py as ko


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(golden_dir / "env_layer.npz")


def test_obs_local_and_global_match_reference(G):
    n = len(G["obs_local"])
    assert n >= 64
    for i in range(n):
        inp = {k: G[k][i] for k in ("palm_xpos", "palm_xmat", "finger_xpos", "obj_xpos", "link7_xpos", "site_xpos",
                                    "sensordata", "obj_size")}
        ol, og = ko.env_obs_from_inputs(inp)
        # fp64 vs fp64: only the acos/pow library calls can differ in the last ulps
        np.testing.assert_allclose(ol, G["obs_local"][i], rtol=1e-12, atol=1e-13, err_msg=f"case {i} local")
        np.testing.assert_allclose(og, G["obs_global"][i], rtol=1e-12, atol=1e-13, err_msg=f"case {i} global")


def test_both_palm_sensor_branches_are_covered(G):
    hit = G["palm_hit"].astype(bool)
    assert hit.any() and (~hit).any()
    assert np.allclose(G["obs_local"][~hit][:, 70:73], 0.2)
    assert not np.allclose(G["obs_local"][hit][:, 70:73], 0.2)


def test_ctrl_mapping_and_palm_transform(G):
    for i in range(len(G["action"])):
        Tfw, wrist, ctrl = ko.env_ctrl(G["palm_xpos"][i], G["palm_xmat"][i], G["action"][i])
        np.testing.assert_allclose(Tfw, G["Tfw"][i], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(wrist, G["wrist"][i], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(ctrl, G["ctrl"][i], rtol=1e-13, atol=1e-15)


def test_reward_done_info(G):
    assert G["done"].any() and not G["done"].all()
    for i in range(len(G["reward"])):
        r, d, info = ko.env_reward(G["obj_xpos"][i][2])
        assert r == G["reward"][i]
        assert d == bool(G["done"][i])
        np.testing.assert_array_equal(info, G["info"][i])


def test_check_grasp_known_answers(G):
    out = [ko.check_grasp(o, n) for o, n in zip(G["cg_old"], G["cg_new"])]
    assert 0 < sum(out) < len(out)
    np.testing.assert_array_equal(np.array(out, dtype=np.float64), G["cg_out"])
