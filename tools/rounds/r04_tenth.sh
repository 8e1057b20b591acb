#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04k
mkdir -p $out
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in default noscan; do
  lib=$GRAFT_REPO_ROOT/kinovagrasping_amd/libkinova_sim.so; [ $v = noscan ] && lib=$GRAFT_REPO_ROOT/kinovagrasping_amd/libkinova_sim_noscan.so
  KS_LIB=$lib python bench.py --no-cpu-baseline --steady-steps 0 > $out/bench_${v}_$rep.log 2>&1
  KS_LIB=$lib python bench.py --no-cpu-baseline --mode sim > $out/bench_sim_${v}_$rep.log 2>&1
  KS_LIB=$lib python bench.py --no-cpu-baseline --rollout lockstep --steady-steps 0 > $out/bench_lock_${v}_$rep.log 2>&1
done; done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_obs_contacts.py tests/test_mujoco_recorded.py -m gpu -q > $out/gputests.log 2>&1; echo "pytest rc $?" >> $out/gputests.log; tail -3 $out/gputests.log
