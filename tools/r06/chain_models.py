#!/usr/bin/env python3
"""CPU analysis of gpurun_out/r06/env_chains.npz (tools/r06/env_chains.py): launch time models for the free-running rollout kernel."""
import numpy as np, sys
d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06/env_chains.npz")
newton, hull, cyc = d["newton"], d["hull"], d["cyc"]
S, n = cyc.shape
wave = cyc.reshape(S, n // 4, 4)
print("cycles of the 4 envs of a wave agree:", np.abs(wave.max(2) - wave.min(2)).max() / wave.mean())
cw = wave[:, :, 0]                                   # [S][1024] the wave's cycles per env-step (stamped build)
# the env-level cost model: wave cycles ~ a + b max Newton + c max hull
A = np.stack([np.ones(S * n // 4), newton.reshape(S, -1, 4).max(2).ravel(), hull.reshape(S, -1, 4).max(2).ravel()], 1)
coef, *_ = np.linalg.lstsq(A, cw.ravel(), rcond=None)
pred = A @ coef
print("wave cycles ~ %.0f + %.0f x max Newton + %.0f x max live hull pairs; residual rms %.3f of mean" % (*coef, np.sqrt(np.mean((pred - cw.ravel()) ** 2)) / cw.mean()))
ce = coef[0] + coef[1] * newton + coef[2] * hull      # [S][n] an env's own cost if it had a wave to itself
for L in (20, 60):
    rows = []
    for s0 in range(0, S - L + 1, L):
        c = cw[s0:s0 + L]
        wg = c.reshape(L, -1, 4).max(2).sum(0)          # workgroup with barriers: sum over steps of its slowest wave
        fw = c.sum(0)                                    # free waves
        env = ce[s0:s0 + L].sum(0)
        # modelled variants on the env-level cost model
        mw = lambda order: np.take_along_axis(ce[s0:s0 + L], order, 1)
        rows.append((wg.max(), wg.mean(), fw.max(), fw.mean(), env.max(), env.mean(), np.sort(fw)[-10:].mean()))
    r = np.mean(rows, 0)
    print(f"launch of {L} env-steps (k cycles per env-step): barriers: slowest workgroup {r[0]/L/1e3:.0f} (mean {r[1]/L/1e3:.0f}); free waves: slowest wave {r[2]/L/1e3:.0f} "
          f"(mean {r[3]/L/1e3:.0f}, mean of the 10 slowest {r[6]/L/1e3:.0f}); an env alone (model): slowest {r[4]/L/1e3:.0f} (mean {r[5]/L/1e3:.0f})")
# how persistent is a wave's cost?  autocorrelation of cw over steps
x = cw - cw.mean(1, keepdims=True)
for lag in (1, 5, 15, 30):
    print(f"lag {lag}: corr {np.mean(x[lag:] * x[:-lag]) / np.mean(x * x):.3f}", end="; ")
print()
# distribution of a wave's chain over a launch of 20
c = cw[:20].sum(0) / 20 / 1e3
print("wave chains over 20 steps, percentiles (k cycles/step):", np.percentile(c, [0, 10, 50, 90, 99, 100]).round(0))
np.save("/tmp/ce.npy", ce)
