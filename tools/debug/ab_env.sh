# A/B of an environment switch on one box: bash tools/debug/ab_env.sh KS_OBS_IN_STEP   (runs VAR=0 and VAR=1 alternately)
v=$1
for val in 0 1 0 1; do
  env $v=$val python bench.py --no-cpu-baseline --steady-updates 600 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v=$val', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['steady_state']['value'], d['steady_state']['k_env_step_avg_launch_ms'], d['nonfinite_envs'])"
done
for val in 0 1; do
  env $v=$val python bench.py --no-cpu-baseline --mode sim 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v=$val sim-only', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
