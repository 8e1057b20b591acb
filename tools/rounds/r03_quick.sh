#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_d
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_obs_contacts.py tests/test_gpu_parity.py -m gpu -q -x -k "one_step or long_horizon or config2 or primitive" > $out/pytest_sel.log 2>&1; tail -3 $out/pytest_sel.log
timeout 900 python3 bench.py --no-cpu-baseline --config 5 > $out/config5.log 2>&1
timeout 300 python3 bench.py --no-cpu-baseline --mode sim > $out/sim.log 2>&1
timeout 600 python3 bench.py --no-cpu-baseline > $out/ddpg.log 2>&1
for f in $out/config5.log $out/sim.log $out/ddpg.log; do echo $f; tail -1 $f | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], (d.get('steady_state') or {}).get('value'), d['status_counts'], d['roofline']['avg_launch_ms'])
except Exception as e: print('ERR', e)"; done
