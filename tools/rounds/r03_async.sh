#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_f
mkdir -p $out
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline "$@" > $out/$name.log 2>&1; tail -1 $out/$name.log | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], (d.get('steady_state') or {}).get('value'), d['status_counts'], d['roofline']['avg_launch_ms'], d['config'].get('free_running'), d['config']['learner_updates_timed'])
except Exception as e: print('$name ERR', e); import subprocess; print(open('$out/$name.log').read()[-1500:])"; }
run free_c10
run lockstep --rollout lockstep
run free_c5 --chunk 5
run free_c20 --chunk 20
run free_c30 --chunk 30 --steps 60
run free_driver --steps 20 --warmup 5
run free_cfg5 --config 5
run lock_cfg5 --config 5 --rollout lockstep
