#!/usr/bin/env python3
"""Dev-only: (1) let the REFERENCE's ReplayBuffer_Queue write a small replay bundle (utils.py:345-365) into
tests/golden/replay_bundle/ - data the loader test reads; (2) check that a bundle written by OUR
save_reference_bundle is read back correctly by the reference's store_saved_data_into_replay (utils.py:367-400).

numpy >= 1.24 refuses np.save of ragged nested lists (legacy numpy pickled them as object arrays, which is what the
reference's loader expects back): the harness wraps np.save with that legacy conversion, nothing else is patched.
"""
import contextlib
import io
import shutil
import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
REF = Path("/root/reference/gym-kinova-gripper")
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REF))


def main():
    import utils as ref_utils
    from kinovagrasping_amd.replay import load_reference_bundle, save_reference_bundle
    rng = np.random.Generator(np.random.PCG64(77))
    buf = ref_utils.ReplayBuffer_Queue(82, 4, max_episode=100, n_steps=5)
    lens = [7, 30, 12]
    eps = []
    for L in lens:
        buf.add_episode(1)
        ep = dict(state=rng.normal(size=(L, 82)), action=rng.uniform(0, 0.8, (L, 4)), next_state=rng.normal(size=(L, 82)),
                  reward=np.where(np.arange(L) == L - 1, 50.0, 0.0), not_done=np.where(np.arange(L) == L - 1, 0.0, 1.0))
        for t in range(L):
            buf.add(ep["state"][t], ep["action"][t], ep["next_state"][t], ep["reward"][t], float(t == L - 1))
        buf.add_episode(0)
        eps.append(ep)
    orig_save = np.save

    def legacy_save(file, arr, **kw):
        if isinstance(arr, list):
            try:
                arr = np.array(arr)
            except ValueError:
                o = np.empty(len(arr), dtype=object)
                for i, r in enumerate(arr):
                    o[i] = r
                arr = o
        return orig_save(file, arr, **kw)

    dst = REPO / "tests" / "golden" / "replay_bundle"
    shutil.rmtree(dst, ignore_errors=True)
    ref_utils.np.save = legacy_save
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            buf.save_replay_buffer(str(dst))
    finally:
        ref_utils.np.save = orig_save
    got, info = load_reference_bundle(dst)
    assert len(got) == 3 and [len(g["reward"]) for g in got] == lens
    for g, e in zip(got, eps):
        for k in e:
            assert np.allclose(g[k], e[k], atol=1e-6), k
    np.savez_compressed(REPO / "tests" / "golden" / "replay_bundle_expected.npz",
                        **{f"ep{i}_{k}": v.astype(np.float32) for i, e in enumerate(eps) for k, v in e.items()}, info=info)
    # our writer -> the reference's reader
    with tempfile.TemporaryDirectory() as td:
        save_reference_bundle(td, eps, max_episode=100)
        buf2 = ref_utils.ReplayBuffer_Queue(82, 4, max_episode=100, n_steps=5)
        with contextlib.redirect_stdout(io.StringIO()):
            buf2.store_saved_data_into_replay(td + "/")
        # reference quirk: its loader drops the trailing open episode with remove_episode(-1), which also decrements
        # replay_ep_num - the same happens with a bundle the reference wrote itself
        buf3 = ref_utils.ReplayBuffer_Queue(82, 4, max_episode=100, n_steps=5)
        with contextlib.redirect_stdout(io.StringIO()):
            buf3.store_saved_data_into_replay(str(dst) + "/")
        assert (buf2.replay_ep_num, buf2.size, len(buf2.state)) == (buf3.replay_ep_num, buf3.size, len(buf3.state)) == (2, sum(lens), 3)
        for i, e in enumerate(eps):
            assert np.allclose(np.array(buf2.state[i], dtype=float), e["state"]) and np.allclose(np.array(buf2.reward[i], dtype=float), e["reward"])
        np.random.seed(0)
        st, ac, ns, rw, nd = buf2.sample_batch_nstep(2)
        assert st.shape[1:] == (5, 82)
    print("wrote", dst, sorted(p.name for p in dst.iterdir()), "info", info, "; reference reads our bundle: ok")


if __name__ == "__main__":
    main()
