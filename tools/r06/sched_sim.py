#!/usr/bin/env python3
"""CPU: event simulation of env -> wave dealing schemes for the free-running rollout kernel on the per-env work counters of tools/r06/env_chains.py.
A wave-step of a group of envs costs a + b max(Newton) + c max(live hull pairs) (fitted to the stamped build's wave cycles)."""
import numpy as np, sys, heapq
d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06/env_chains.npz")
newton, hull, cyc = d["newton"], d["hull"], d["cyc"]
S, n = cyc.shape
cw = cyc.reshape(S, n // 4, 4)[:, :, 0]
A = np.stack([np.ones(S * n // 4), newton.reshape(S, -1, 4).max(2).ravel(), hull.reshape(S, -1, 4).max(2).ravel()], 1)
coef, *_ = np.linalg.lstsq(A, cw.ravel(), rcond=None)
a, b, c = coef
def gcost(envs, steps):
    return a + b * max(newton[s, e] for e, s in zip(envs, steps)) + c * max(hull[s, e] for e, s in zip(envs, steps))
def key_of(e, s):            # what a kernel knows: the env's previous env-step
    return 0.0 if s == 0 else b * newton[s - 1, e] + c * hull[s - 1, e]
def simulate(L, s0, scheme, nw=1024, scope=None, nclass=1, overhead=0.0):
    """scheme 'static': env e in wave e // 4 forever.  'fifo': a wave pops 4 ready envs.  'sorted': ready envs kept in nclass FIFOs by cost key.
    scope = envs per pool (None: global; 16: the workgroup's own envs, 4 waves per pool)"""
    pools = [list(range(n))] if scope is None else [list(range(i, i + scope)) for i in range(0, n, scope)]
    wpp = nw // len(pools)
    finish = 0.0; busy = 0.0
    for pool in pools:
        step = {e: 0 for e in pool}
        if scheme == 'static':
            for w in range(wpp):
                envs = pool[4 * w:4 * w + 4]
                t = sum(gcost(envs, [s0 + s] * 4) for s in range(L))
                finish = max(finish, t); busy += t
            continue
        ready = [[] for _ in range(nclass)]      # FIFOs of (time available, env)
        for e in pool: ready[0].append((0.0, e))
        waves = [(0.0, w) for w in range(wpp)]   # (time free, id)
        heapq.heapify(waves)
        remaining = len(pool) * L
        pending = []                              # (time, env) pushed back later (heap)
        while remaining > 0:
            tw, w = heapq.heappop(waves)
            # move pending pushes that have happened by tw into the FIFOs
            while pending and pending[0][0] <= tw:
                tp, e = heapq.heappop(pending)
                k = 0
                if nclass > 1:
                    k = min(nclass - 1, int(key_of(e, s0 + step[e]) / kmax * nclass))
                ready[k].append((tp, e))
            avail = sum(len(r) for r in ready)
            if avail == 0:
                # wait for the next push
                tp = pending[0][0]
                heapq.heappush(waves, (tp, w))
                continue
            # take up to 4 from the fullest-first class order: highest class first (expensive envs start early)
            envs = []
            for k in range(nclass - 1, -1, -1):
                while ready[k] and len(envs) < 4:
                    envs.append(ready[k].pop(0)[1])
                if len(envs) == 4: break
            cost = gcost(envs, [s0 + step[e] for e in envs]) + overhead
            t1 = tw + cost
            busy += cost
            for e in envs:
                step[e] += 1; remaining -= 1
                if step[e] < L: heapq.heappush(pending, (t1, e))
            finish = max(finish, t1)
            heapq.heappush(waves, (t1, w))
    return finish / L / 1e3, busy / nw / L / 1e3
kmax = (b * newton + c * hull).max() * 0.6
for L in (20, 60):
    for s0 in (0, 60):
        if s0 + L > S: continue
        print(f"L={L} s0={s0}: static {simulate(L, s0, 'static')}, fifo-global {simulate(L, s0, 'fifo')}, fifo-wg {simulate(L, s0, 'fifo', scope=16)}, "
              f"sorted8-global {simulate(L, s0, 'sorted', nclass=8)}, sorted4-wg64 {simulate(L, s0, 'sorted', scope=64, nclass=4)}, sorted8-global+50k {simulate(L, s0, 'sorted', nclass=8, overhead=50e3)}")
