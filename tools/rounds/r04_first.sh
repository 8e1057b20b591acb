#!/bin/bash
# round 4, first measurement batch after the margin / mesh-frame fix (one box)
out=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $out
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -s > $out/gputests.log 2>&1; echo "pytest rc $?" >> $out/gputests.log
python tools/debug/fp64_first_diff.py CylinderB 6 > $out/fp64_diff_CylinderB.txt 2>&1
python tools/debug/fp64_first_diff.py Vase1B 4 > $out/fp64_diff_Vase1B.txt 2>&1
python -m tests.studies.long_horizon > $out/long_horizon.txt 2>&1
KS_LIB=$GRAFT_REPO_ROOT/kinovagrasping_amd/libkinova_sim_nowarm.so python -m tests.studies.long_horizon > $out/long_horizon_mpr_cold.txt 2>&1
for v in default nowarm; do
  lib=$GRAFT_REPO_ROOT/kinovagrasping_amd/libkinova_sim.so; [ $v = nowarm ] && lib=$GRAFT_REPO_ROOT/kinovagrasping_amd/libkinova_sim_nowarm.so
  KS_LIB=$lib python bench.py --no-cpu-baseline > $out/bench_$v.log 2>&1
  KS_LIB=$lib python bench.py --no-cpu-baseline --mode sim > $out/bench_sim_$v.log 2>&1
done
tail -3 $out/gputests.log
