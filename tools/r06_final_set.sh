#!/bin/bash
# final measurement set of round 6 (usage: tools/r05_final_set.sh [tag]): bench lines, kernel stats + the per-env-step join, k_rollout counters
tag=${1:-r06_a}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
python3 $B > $out/ddpg_bench.log 2>&1
python3 $B --steps 20 --warmup 5 --no-cpu-baseline > $out/ddpg_driver_form_bench.log 2>&1
python3 $B --rollout lockstep --no-cpu-baseline > $out/ddpg_lockstep_bench.log 2>&1
python3 $B --mode sim --no-cpu-baseline > $out/sim_bench.log 2>&1
python3 $B --config 5 --no-cpu-baseline > $out/config5_bench.log 2>&1
python3 $B --config 5 --cohort 16 --no-cpu-baseline > $out/config5_cohort16_bench.log 2>&1
python3 $B --expert-prob 0.3 --no-cpu-baseline > $out/ddpgfd_bench.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ddpg -o ddpg -- python3 $B --no-cpu-baseline > $out/ddpg_prof_bench.log 2>&1
cp /tmp/prof_ddpg/ddpg_kernel_stats.csv $out/ddpg_kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/rollout_trace_join.py /tmp/prof_ddpg/ddpg_kernel_trace.csv $out/ddpg_prof_bench.log > $out/ddpg_k_rollout_per_step.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lock -o lock -- python3 $B --rollout lockstep --no-cpu-baseline > $out/ddpg_lockstep_prof_bench.log 2>&1
cp /tmp/prof_lock/lock_kernel_stats.csv $out/ddpg_lockstep_kernel_stats.csv
cd $GRAFT_REPO_ROOT
timeout 1200 bash tools/pmc_run.sh free gpurun_out/pmc_free > $out/pmc_free_summary.txt 2>&1
for f in ddpg ddpg_driver_form ddpg_lockstep sim config5 config5_cohort16 ddpgfd; do grep '^{' $out/${f}_bench.log | tail -1 | cut -c1-150; done
head -4 $out/ddpg_k_rollout_per_step.txt; tail -4 $out/pmc_free_summary.txt
timeout 1200 bash tools/pmc_run.sh sim gpurun_out/pmc_sim > $out/pmc_sim_summary.txt 2>&1
timeout 1500 bash tools/pmc_run.sh ddpg gpurun_out/pmc_ddpg > $out/pmc_ddpg_summary.txt 2>&1
python3 $B --no-cpu-baseline --init-policy none > $out/ddpg_noinit_bench.log 2>&1
KS_ROLLOUT_WAVES=0 python3 $B --no-cpu-baseline > $out/ddpg_waves0_bench.log 2>&1
for v in r5warm sm0; do
  if [ -f $GRAFT_REPO_ROOT/tools/experiments/build/libkinova_sim_$v.so ]; then
    KS_LIB=$GRAFT_REPO_ROOT/tools/experiments/build/libkinova_sim_$v.so python3 $B --no-cpu-baseline > $out/ddpg_${v}_bench.log 2>&1
    KS_LIB=$GRAFT_REPO_ROOT/tools/experiments/build/libkinova_sim_$v.so python3 $B --mode sim --no-cpu-baseline > $out/sim_${v}_bench.log 2>&1
  fi
done
python3 $B --no-cpu-baseline --config 5 --cohort 16 --rollout lockstep > $out/config5_cohort16_lockstep_bench.log 2>&1
KS_LEARNER_FORK=1 python3 $B --no-cpu-baseline --rollout lockstep > $out/ddpg_lockstep_fork_bench.log 2>&1
KS_LEARNER_FORK=1 python3 $B --no-cpu-baseline --rollout lockstep --serial-learner > $out/ddpg_lockstep_serial_fork_bench.log 2>&1
python3 $B --no-cpu-baseline --rollout lockstep --serial-learner > $out/ddpg_lockstep_serial_bench.log 2>&1
for f in ddpg_noinit ddpg_waves0 ddpg_r5warm sim_r5warm ddpg_sm0 sim_sm0 config5_cohort16_lockstep ddpg_lockstep_fork ddpg_lockstep_serial_fork ddpg_lockstep_serial; do grep '^{' $out/${f}_bench.log | tail -1 | cut -c1-150; done
