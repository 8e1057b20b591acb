#!/bin/bash
# round 5, first measurement batch (gpurun): bench lines after the operand-order change, fp64 rate, long-horizon study, DDPGfD in the reference's regime
out=$GRAFT_REPO_ROOT/gpurun_out/r05_a
mkdir -p $out
cd $GRAFT_REPO_ROOT
B=bench.py
python3 $B > $out/ddpg_bench.log 2>&1
python3 $B --mode sim --no-cpu-baseline > $out/sim_bench.log 2>&1
python3 $B --config 5 --no-cpu-baseline > $out/config5_bench.log 2>&1
python3 $B --config 5 --rollout free --no-cpu-baseline > $out/config5_free_bench.log 2>&1
python3 $B --config 5 --cohort 16 --no-cpu-baseline > $out/config5_cohort16_bench.log 2>&1
for f in ddpg sim config5 config5_free config5_cohort16; do grep '^{' $out/${f}_bench.log | tail -1 | cut -c1-160; done
python3 tools/r05/fp64_rate.py > $out/fp64_rate.txt 2>&1; cat $out/fp64_rate.txt
python3 -m tests.studies.long_horizon > $out/long_horizon.txt 2>&1; tail -3 $out/long_horizon.txt
# DDPGfD with the REFERENCE's schedule: ~3.3 updates per stored transition (100 train_batch calls per 30-step episode of one env, main_DDPGfD.py:474-486),
# tau 0.0005 every 10th update (DDPGfD.py:54, 360-366), 32 envs -> 107 updates per env-step of the batch
timeout 1500 python3 examples/train_ddpgfd.py --envs 32 --steps 3000 --updates-per-step 107 --tau 0.0005 --target-every 10 --eval-every 300 > $out/train_ref_regime_32.log 2>&1
grep -i "eval\|success" $out/train_ref_regime_32.log | tail -12
