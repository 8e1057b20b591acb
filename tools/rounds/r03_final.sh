#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_z
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/driver_form.log 2>&1
python3 bench.py > $out/default.log 2>&1
python3 bench.py --rollout free --no-cpu-baseline > $out/free.log 2>&1
tail -3 $out/pytest_gpu.log; tail -1 $out/smoke.log; grep real $out/driver_form.log
for f in driver_form default free; do grep '^{' $out/$f.log | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'], d['steady_state']['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['status_counts'], (d.get('cpu_baseline') or {}).get('value'))"; done
