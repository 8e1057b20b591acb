# bash tools/debug/ab_lib_env.sh NAME...   (libkinova_sim_NAME.so variants; ddpg + steady + sim-only)
for n in "$@"; do
  L=$PWD/kinovagrasping_amd/libkinova_sim_$n.so
  KS_LIB=$L python bench.py --no-cpu-baseline --steady-updates 600 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['steady_state']['value'], d['steady_state']['k_env_step_avg_launch_ms'], d['nonfinite_envs'])"
  KS_LIB=$L python bench.py --no-cpu-baseline --mode sim 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n sim-only', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
