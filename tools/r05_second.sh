#!/bin/bash
# round 5, second batch: the lines the ready queue changed + the multi-geom build's counters + the mixed stage context
out=$GRAFT_REPO_ROOT/gpurun_out/r05_b
mkdir -p $out
cd $GRAFT_REPO_ROOT
B=bench.py
python3 $B --config 5 --no-cpu-baseline > $out/config5_bench.log 2>&1
python3 $B --config 5 --cohort 16 --no-cpu-baseline > $out/config5_cohort16_bench.log 2>&1
python3 $B --config 5 --rollout lockstep --no-cpu-baseline > $out/config5_lockstep_bench.log 2>&1
KS_ROLLOUT_DEAL=rr python3 $B --config 5 --rollout free --no-cpu-baseline > $out/config5_rr_bench.log 2>&1
python3 $B --no-cpu-baseline > $out/ddpg_bench.log 2>&1
for f in config5 config5_cohort16 config5_lockstep config5_rr ddpg; do grep '^{' $out/${f}_bench.log | tail -1 | cut -c1-150; done
python3 tools/debug/mg_stage_rate.py > $out/mg_stage_rate.txt 2>&1; tail -2 $out/mg_stage_rate.txt
python3 tools/debug/mg_perf.py > $out/mg_perf.txt 2>&1; tail -6 $out/mg_perf.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 bash tools/pmc_run.sh mg gpurun_out/pmc_mg > $out/pmc_mg_summary.txt 2>&1; tail -3 $out/pmc_mg_summary.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -o c5 -- python3 $B --config 5 --no-cpu-baseline > $out/config5_prof_bench.log 2>&1
cp /tmp/prof_c5/c5_kernel_stats.csv $out/config5_kernel_stats.csv; head -3 $out/config5_kernel_stats.csv | cut -c1-200
