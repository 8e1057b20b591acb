#!/bin/bash
# round 6: the two-lane penetration query against the one-lane build (tools/experiments/build/libkinova_sim_nosplit.so = -DKS_MPR_SPLIT=0): parity, then rates
mkdir -p gpurun_out/r06s2
o=gpurun_out/r06s2/ab_split.txt; : > $o
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long_horizon or env_step_path or one_step or config1 or config2" > gpurun_out/r06s2/par_split.txt 2>&1; tail -5 gpurun_out/r06s2/par_split.txt >> $o
timeout 900 python -m pytest tests/test_gpu_async.py -m gpu -x -q -k "scheduling or lock_step" > gpurun_out/r06s2/async_split.txt 2>&1; tail -3 gpurun_out/r06s2/async_split.txt >> $o
VARIANTS="nosplit default nosplit default" bash tools/r06/rate_variants.sh >> $o 2>&1
cat $o
