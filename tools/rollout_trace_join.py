#!/usr/bin/env python3
"""rocprofv3 kernel trace of a free-running bench run x the bench line's own list of launch sizes -> k_rollout duration per env-step.
k_rollout launches advance different numbers of env-steps (priming 36, pre-training 60 + 40, warm-up, the timed launch(es), the steady window),
so `--stats`' per-launch average is not the per-env-step figure `roofline.avg_launch_ms` reports; this joins the two.
usage: tools/rollout_trace_join.py <kernel_trace.csv> <bench.log>"""
import csv
import json
import sys

trace, log = sys.argv[1], sys.argv[2]
line = [l for l in open(log) if l.startswith("{")][-1]
d = json.loads(line)
fr = d["config"]["free_running"]
steps = fr["launch_steps"]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(trace)) if "k_rollout" in r["Kernel_Name"])
if len(rows) != len(steps):
    print(f"warning: {len(rows)} k_rollout dispatches in the trace, {len(steps)} launches in the bench line")
n = min(len(rows), len(steps))
dur = [(rows[i][1] - rows[i][0]) / 1e6 for i in range(n)]
print(f"k_rollout: {n} launches, {sum(steps[:n])} env-steps, {sum(dur):.2f} ms in total = {sum(dur) / sum(steps[:n]):.4f} ms per env-step over the whole run "
      f"(--stats average per LAUNCH: {sum(dur) / n:.3f} ms)")
f0, k = fr["first_timed_launch"], fr["timed_launches"]
td, ts = sum(dur[f0:f0 + k]), sum(steps[f0:f0 + k])
print(f"timed region: launches {f0}..{f0 + k - 1} ({'+'.join(map(str, steps[f0:f0 + k]))} env-steps): {td:.3f} ms = {td / ts:.4f} ms per env-step by the trace; "
      f"bench line roofline.avg_launch_ms {d['roofline']['avg_launch_ms']} (HIP events on the launch stream), ms_per_step {d['ms_per_step']}")
sd = dur[f0 + k:]
if sd:
    print(f"steady window: {len(sd)} launches, {sum(sd) / sum(steps[f0 + k:n]):.4f} ms per env-step by the trace; bench line {d['steady_state']['k_env_step_avg_launch_ms']}")
for i in range(n):
    tag = "timed" if f0 <= i < f0 + k else ("steady" if i >= f0 + k else "")
    print(f"  launch {i:3d}: {steps[i]:3d} env-steps  {dur[i]:9.3f} ms  {dur[i] / steps[i]:.4f} ms per env-step  {tag}")
