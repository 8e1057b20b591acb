#!/bin/bash
# round-3 second GPU call: GPU suite, long-horizon parity study, DDPGfD (expert mix) and config-5 bench lines
out=$GRAFT_REPO_ROOT/gpurun_out/r03_e
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -m gpu -q -s > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log
timeout 1200 python3 -m tests.studies.long_horizon > $out/long_horizon.txt 2> $out/long_horizon.err
timeout 600 python3 bench.py --no-cpu-baseline --expert-prob 0.3 > $out/ddpgfd_expert.log 2>&1
timeout 600 python3 bench.py --no-cpu-baseline > $out/ddpg.log 2>&1
timeout 900 python3 bench.py --no-cpu-baseline --config 5 > $out/config5.log 2>&1
grep -E "passed|failed|error" $out/pytest_gpu.log | tail -5
grep -E "^FAILED|Error" $out/pytest_gpu.log | head -20
head -30 $out/long_horizon.txt; tail -3 $out/long_horizon.txt; tail -3 $out/long_horizon.err
for f in $out/ddpgfd_expert.log $out/ddpg.log $out/config5.log; do echo $f; tail -1 $f | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], (d.get('steady_state') or {}).get('value'), d['status_counts'], d['roofline']['avg_launch_ms'], d['config'].get('expert_mix'))
except Exception as e: print('ERR', e)"; done
