#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04i
mkdir -p $out
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --config 5 > $out/bench_cfg5_free_runs.log 2>&1
KS_ROLLOUT_DEAL=rr python bench.py --no-cpu-baseline --config 5 > $out/bench_cfg5_free_rr.log 2>&1
python bench.py --no-cpu-baseline --config 5 --rollout lockstep > $out/bench_cfg5_lock.log 2>&1
for i in 1 2 3; do python bench.py --no-cpu-baseline > $out/bench_default_$i.log 2>&1; done
python bench.py --no-cpu-baseline --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.log 2>&1
python bench.py --no-cpu-baseline --rollout lockstep > $out/bench_lockstep.log 2>&1
python bench.py --no-cpu-baseline --init-policy none > $out/bench_default_noinit.log 2>&1
python -m pytest tests -m gpu -q > $out/gputests.log 2>&1; echo "pytest rc $?" >> $out/gputests.log
tail -3 $out/gputests.log
