#!/bin/bash
# experiment (round 2): ray pool on / off (KS_RAY_POOL), alternating runs of the bench on one box
for val in 0 unset 0 unset; do
  if [ $val = unset ]; then unset KS_RAY_POOL; else export KS_RAY_POOL=$val; fi
  python bench.py --no-cpu-baseline --steady-updates 600 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pool=$val', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['steady_state']['value'], d['steady_state']['k_env_step_avg_launch_ms'], d['nonfinite_envs'])"
done
for val in 0 unset; do
  if [ $val = unset ]; then unset KS_RAY_POOL; else export KS_RAY_POOL=$val; fi
  python bench.py --no-cpu-baseline --mode sim 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pool=$val sim-only', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
