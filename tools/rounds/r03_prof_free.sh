#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py > $out/ddpg_bench.log 2>&1
python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/ddpg_driver_form_bench.log 2>&1
python3 $GRAFT_REPO_ROOT/bench.py --rollout lockstep --no-cpu-baseline > $out/ddpg_lockstep_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_free -o free -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $out/ddpg_prof_bench.log 2>&1
cp /tmp/prof_free/free_kernel_stats.csv $out/ddpg_kernel_stats.csv
cd $GRAFT_REPO_ROOT
timeout 1500 bash tools/pmc_run.sh free gpurun_out/pmc_free > $out/pmc_free_summary.txt 2>&1
for f in ddpg ddpg_driver_form ddpg_lockstep; do grep '^{' $out/${f}_bench.log | tail -1 | cut -c1-160; done; tail -12 $out/pmc_free_summary.txt
