#!/bin/bash
# round 4 soak runs: long timed regions, status counters (contact overflow, non-finite state, ray time-out, Newton cap), dropped episodes
out=$GRAFT_REPO_ROOT/gpurun_out/${SOAK_TAG:-r04_soak}
mkdir -p $out
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 1500 python3 bench.py --no-cpu-baseline "$@" > $out/$name.log 2>&1; tail -1 $out/$name.log | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=(d.get('timed_window') or {}).get('regime') or {}
    print('$name', 'env-steps/s', d['value'], 'ms/step', d['ms_per_step'], 'env-steps timed', d['steps']*d['config']['envs_per_gpu'], 'status', d['status_counts'], 'free-running', {k:v for k,v in (d['config'].get('free_running') or {}).items() if k in ('episodes_finished','episodes_dropped','pacing_timeouts')}, 'regime', r)
except Exception as e: print('$name ERR', e); print(open('$out/$name.log').read()[-1500:])"; }
run long_free --steps 3000 --steady-steps 0
run long_lock --steps 3000 --steady-steps 0 --rollout lockstep
run long_cfg5 --steps 1500 --steady-steps 0 --config 5
run long_cfg5_lock --steps 1000 --steady-steps 0 --config 5 --rollout lockstep
run long_sim --steps 3000 --mode sim
run long_noinit --steps 3000 --steady-steps 0 --init-policy none
