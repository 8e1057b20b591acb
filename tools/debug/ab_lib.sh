#!/bin/bash
# A/B of library variants on the default bench (free-running): bash tools/debug/ab_lib.sh NAME [NAME ...]   ("" = the product library)
cd $GRAFT_REPO_ROOT
show() { grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], (d.get('steady_state') or {}).get('value'), d['config']['free_running']['lifted'] if d['config'].get('free_running') else '')"; }
for rep in 1 2; do
for lib in default "$@"; do
  if [ "$lib" != default ]; then export KS_LIB=$PWD/kinovagrasping_amd/libkinova_sim_$lib.so; else unset KS_LIB; fi
  echo -n "lib=$lib: "; python3 bench.py --no-cpu-baseline $BENCH_ARGS 2>/dev/null | show
done; done
