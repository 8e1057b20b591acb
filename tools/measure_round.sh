#!/bin/bash
# Dev tool: the measurements the docs and profiles/ quote, in one GPU call.  Usage: tools/measure_round.sh <tag>
tag=${1:-r03_a}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py > $out/ddpg_bench.log 2>&1
python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/ddpg_driver_form_bench.log 2>&1
python3 $GRAFT_REPO_ROOT/bench.py --mode sim --no-cpu-baseline > $out/sim_bench.log 2>&1
python3 $GRAFT_REPO_ROOT/bench.py --config 5 --no-cpu-baseline > $out/config5_bench.log 2>&1
python3 $GRAFT_REPO_ROOT/bench.py --expert-prob 0.3 --no-cpu-baseline > $out/ddpgfd_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ddpg -o ddpg -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $out/ddpg_prof_bench.log 2>&1
cp /tmp/prof_ddpg/ddpg_kernel_stats.csv $out/ddpg_kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/rollout_trace_join.py /tmp/prof_ddpg/ddpg_kernel_trace.csv $out/ddpg_prof_bench.log > $out/ddpg_k_rollout_per_step.txt 2>&1
python3 $GRAFT_REPO_ROOT/bench.py --rollout lockstep --no-cpu-baseline > $out/ddpg_lockstep_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lock -o lock -- python3 $GRAFT_REPO_ROOT/bench.py --rollout lockstep --no-cpu-baseline > $out/ddpg_lockstep_prof_bench.log 2>&1
cp /tmp/prof_lock/lock_kernel_stats.csv $out/ddpg_lockstep_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sim -o sim -- python3 $GRAFT_REPO_ROOT/bench.py --mode sim --no-cpu-baseline > $out/sim_prof_bench.log 2>&1
cp /tmp/prof_sim/sim_kernel_stats.csv $out/sim_kernel_stats.csv
cd $GRAFT_REPO_ROOT
bash tools/pmc_run.sh sim gpurun_out/pmc_sim > $out/pmc_sim_summary.txt 2>&1
bash tools/pmc_run.sh ddpg gpurun_out/pmc_ddpg > $out/pmc_ddpg_summary.txt 2>&1
timeout 900 bash tools/pmc_run.sh free gpurun_out/pmc_free > $out/pmc_free_summary.txt 2>&1
for f in ddpg ddpg_driver_form ddpg_lockstep sim config5 ddpgfd; do tail -1 $out/${f}_bench.log | cut -c1-160; done
