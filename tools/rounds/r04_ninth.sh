#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04j
mkdir -p $out
cd $GRAFT_REPO_ROOT
KS_INIT_POLICY=kinovagrasping_amd/assets/bench_policy/ddpg_256_256 KS_LIB=$GRAFT_REPO_ROOT/kinovagrasping_amd/libkinova_sim_stamp.so python tools/gpu_wgtime.py policy-train:300 > $out/wgtime_trained.txt 2>&1
KS_LIB=$GRAFT_REPO_ROOT/kinovagrasping_amd/libkinova_sim_stamp.so python tools/gpu_wgtime.py policy-train:900 > $out/wgtime_untrained900.txt 2>&1
python bench.py --no-cpu-baseline --config 5 > $out/bench_cfg5.log 2>&1
tail -12 $out/wgtime_trained.txt | cut -c1-900
