#!/bin/bash
# Dev tool: the measurements the docs and profiles/ quote, in one GPU call.  Usage: tools/measure_round.sh <tag>
tag=${1:-r02_h}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py > $out/ddpg_bench.log 2>&1
python3 $GRAFT_REPO_ROOT/bench.py --mode sim --no-cpu-baseline > $out/sim_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ddpg -o ddpg -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $out/ddpg_prof_bench.log 2>&1
cp /tmp/prof_ddpg/ddpg_kernel_stats.csv $out/ddpg_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sim -o sim -- python3 $GRAFT_REPO_ROOT/bench.py --mode sim --no-cpu-baseline > $out/sim_prof_bench.log 2>&1
cp /tmp/prof_sim/sim_kernel_stats.csv $out/sim_kernel_stats.csv
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_r2; mkdir -p gpurun_out/pmc_r2
bash tools/pmc_run.sh > $out/pmc_summary.txt 2>&1
rm -rf gpurun_out/pmc_r2/p*/
tail -1 $out/ddpg_bench.log | cut -c1-200
tail -1 $out/sim_bench.log | cut -c1-200
