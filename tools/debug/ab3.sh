#!/bin/bash
# same-box A/B of library variants: bash tools/debug/ab3.sh BASE TIE OPEN NEW   (NEW = the default library)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for n in "$@"; do
  if [ $n = NEW ]; then L=$PWD/kinovagrasping_amd/libkinova_sim.so; else L=$PWD/kinovagrasping_amd/libkinova_sim_$n.so; fi
  KS_LIB=$L python3 bench.py --no-cpu-baseline --steady-steps 150 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n ddpg', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['steady_state']['value'])"
  KS_LIB=$L python3 bench.py --no-cpu-baseline --mode sim 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n sim ', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  if [ $rep = 1 ]; then
  KS_LIB=$L python3 bench.py --no-cpu-baseline --config 5 --steady-steps 150 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n cfg5', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['steady_state']['value'])"
  fi
done
done
