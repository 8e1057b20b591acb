#!/bin/bash
# round-3 first GPU call: GPU suite + bench at three Newton caps
out=$GRAFT_REPO_ROOT/gpurun_out/r03_a
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -q -x -s > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log
for it in 6 12 20; do
  timeout 600 python3 bench.py --no-cpu-baseline --solver-iterations $it > $out/ddpg_it$it.log 2>&1
  timeout 300 python3 bench.py --mode sim --no-cpu-baseline --solver-iterations $it > $out/sim_it$it.log 2>&1
done
timeout 600 python3 bench.py --steps 20 --warmup 5 > $out/driver_form.log 2>&1
tail -5 $out/pytest_gpu.log
for f in $out/ddpg_it*.log $out/sim_it*.log $out/driver_form.log; do echo $f; tail -1 $f | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], (d.get('steady_state') or {}).get('value'), d['status_counts'], d['roofline']['avg_launch_ms'])
except Exception as e: print('ERR', e)"; done
