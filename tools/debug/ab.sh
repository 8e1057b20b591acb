# A/B of library variants on one box: bash tools/debug/ab.sh HEAD NEW ...   (kinovagrasping_amd/libkinova_sim_<name>.so; NEW = the default library)
for n in "$@"; do
  if [ $n = NEW ]; then L=$PWD/kinovagrasping_amd/libkinova_sim.so; else L=$PWD/kinovagrasping_amd/libkinova_sim_$n.so; fi
  KS_LIB=$L python bench.py --no-cpu-baseline --steady-updates 600 2>/dev/null | tail -1 > gpurun_out/bench_$n.json
  python -c "
import json; d=json.load(open('gpurun_out/bench_$n.json')); print('$n', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['steady_state']['value'], d['steady_state']['k_env_step_avg_launch_ms'], d['nonfinite_envs'])"
done
# sim-only leg
for n in "$@"; do
  if [ $n = NEW ]; then L=$PWD/kinovagrasping_amd/libkinova_sim.so; else L=$PWD/kinovagrasping_amd/libkinova_sim_$n.so; fi
  KS_LIB=$L python bench.py --no-cpu-baseline --mode sim 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n sim-only', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
