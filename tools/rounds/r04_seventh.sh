#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04h
mkdir -p $out
cd $GRAFT_REPO_ROOT
python tools/debug/overlap_probe.py 4096 mixed 3 > $out/overlap_4096_mixed3.txt 2>&1
python tools/debug/cfg5_drops.py 8192 mixed 16 14 > $out/drops_8192_mixed_c16.txt 2>&1
python -m pytest tests/test_learner_golden.py tests/test_gpu_learner_state.py tests/test_gpu_async.py tests/test_bench_launch.py -m gpu -q > $out/gputests_learner.log 2>&1; echo "pytest rc $?" >> $out/gputests_learner.log
python bench.py --no-cpu-baseline --config 5 > $out/bench_cfg5_free.log 2>&1
python bench.py --no-cpu-baseline --config 5 --rollout lockstep > $out/bench_cfg5_lock.log 2>&1
python bench.py --no-cpu-baseline > $out/bench_default.log 2>&1
grep "done" $out/overlap_4096_mixed3.txt; grep "^launch" $out/drops_8192_mixed_c16.txt | tail -3; tail -2 $out/gputests_learner.log
bash tools/rounds/r04_sixth.sh
