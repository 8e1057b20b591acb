#!/usr/bin/env python3
"""CPU: what would dealing envs to waves BY EPISODE PHASE at every launch boundary gain?  Wave-step cost model fitted to the stamped build's wave cycles
(a + b max Newton + c max live hull pairs over the wave's four envs) on the per-env counters of tools/r06/env_chains.py."""
import numpy as np, sys
d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06/env_chains_cold.npz")
newton, hull, cyc, t = d["newton"], d["hull"], d["cyc"], d["t"]
S, n = cyc.shape
cw = cyc.reshape(S, n // 4, 4)[:, :, 0]
A = np.stack([np.ones(S * n // 4), newton.reshape(S, -1, 4).max(2).ravel(), hull.reshape(S, -1, 4).max(2).ravel()], 1)
(a, b, c), *_ = np.linalg.lstsq(A, cw.ravel(), rcond=None)
def chains(order, s0, L):
    nw = np.take_along_axis(newton[s0:s0 + L], np.tile(order, (L, 1)), 1).reshape(L, -1, 4).max(2)
    hw = np.take_along_axis(hull[s0:s0 + L], np.tile(order, (L, 1)), 1).reshape(L, -1, 4).max(2)
    return (a + b * nw + c * hw).sum(0) / L / 1e3
print("t at the start of a launch: histogram", np.bincount(t[0].astype(int), minlength=31)[:31])
for L in (20, 60):
    rows = []
    for s0 in range(1, S - L + 1, L):
        ident = np.arange(n)
        # what the kernel knows at the launch boundary: the envs' episode step counters (t after the previous env-step) and their last env-step's own work
        by_t = np.argsort(t[s0 - 1], kind="stable")
        by_cost = np.argsort(b * newton[s0 - 1] + c * hull[s0 - 1], kind="stable")
        by_t_wg = np.concatenate([g[np.argsort(t[s0 - 1][g], kind="stable")] for g in ident.reshape(-1, 16)])       # within the workgroup's 16 envs only
        r = []
        for o in (ident, by_t, by_cost, by_t_wg):
            ch = chains(o, s0, L)
            r += [ch.max(), ch.mean()]
        rows.append(r)
    r = np.mean(rows, 0)
    print(f"launch of {L}: slowest / mean wave chain (k cycles per env-step): as dealt {r[0]:.0f} / {r[1]:.0f}; sorted by episode step {r[2]:.0f} / {r[3]:.0f}; "
          f"sorted by the last env-step's work {r[4]:.0f} / {r[5]:.0f}; by episode step within the workgroup {r[6]:.0f} / {r[7]:.0f}")
