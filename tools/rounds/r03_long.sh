#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_h
mkdir -p $out
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 900 python3 bench.py --no-cpu-baseline "$@" > $out/$name.log 2>&1; tail -1 $out/$name.log | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], (d.get('steady_state') or {}).get('value'), d['status_counts'], d['roofline']['avg_launch_ms'], d['config'].get('free_running'))
except Exception as e: print('$name ERR', e); print(open('$out/$name.log').read()[-1500:])"; }
run long_lock --steps 3000 --steady-steps 0
run long_free --steps 3000 --steady-steps 0 --rollout free
run long_cfg5 --steps 1500 --steady-steps 0 --config 5
run long_expert --steps 1500 --steady-steps 0 --expert-prob 0.3
run free_400_300 --rollout free --hidden 400 300 --steps 60
run lock_400_300 --hidden 400 300 --steps 60
