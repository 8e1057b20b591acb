#!/bin/bash
# experiment (round 2): bench with build variants kinovagrasping_amd/libkinova_sim_<NAME>.so (tools/debug/build_variant.sh) against the product library
for n in SC SD ALL SOLVER; do
  KS_LIB=$PWD/kinovagrasping_amd/libkinova_sim_$n.so python bench.py --no-cpu-baseline --steady-updates 600 2>/dev/null > gpurun_out/bench_$n.json
  python -c "
import json; d=json.load(open('gpurun_out/bench_$n.json')); print('$n', d['value'], d['roofline']['avg_launch_ms'], d['steady_state']['value'], d['steady_state']['k_env_step_avg_launch_ms'], d['nonfinite_envs'])"
done
python bench.py --no-cpu-baseline --steady-updates 600 2>/dev/null > gpurun_out/bench_base.json
python -c "
import json; d=json.load(open('gpurun_out/bench_base.json')); print('default', d['value'], d['roofline']['avg_launch_ms'], d['steady_state']['value'], d['steady_state']['k_env_step_avg_launch_ms'], d['nonfinite_envs'])"
