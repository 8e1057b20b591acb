#!/bin/bash
# round 4, third batch: MPR-first A/B, training sweeps (batch size / updates per step / expert mix)
out=$GRAFT_REPO_ROOT/gpurun_out/r04d
mkdir -p $out
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -s > $out/gputests.log 2>&1; echo "pytest rc $?" >> $out/gputests.log
for v in default nofirst; do
  lib=$GRAFT_REPO_ROOT/kinovagrasping_amd/libkinova_sim.so; [ $v = nofirst ] && lib=$GRAFT_REPO_ROOT/kinovagrasping_amd/libkinova_sim_nofirst.so
  KS_LIB=$lib python bench.py --no-cpu-baseline > $out/bench_$v.log 2>&1
  KS_LIB=$lib python bench.py --no-cpu-baseline --mode sim > $out/bench_sim_$v.log 2>&1
  KS_LIB=$lib python bench.py --no-cpu-baseline --rollout lockstep > $out/bench_lockstep_$v.log 2>&1
  KS_LIB=$lib python bench.py --no-cpu-baseline --config 5 > $out/bench_cfg5_$v.log 2>&1
done
T="python examples/train_ddpgfd.py --envs 4096 --steps 6000 --free-running"
$T --expert-prob 0 > $out/train_plain_b64.log 2>&1
$T --expert-prob 0 --batch-episodes 512 > $out/train_plain_b512.log 2>&1
$T --expert-prob 0.3 --batch-episodes 512 > $out/train_fd_b512.log 2>&1
$T --expert-prob 0.3 --batch-episodes 512 --actor-lr 3e-4 --critic-lr 3e-3 > $out/train_fd_b512_lr3.log 2>&1
$T --expert-prob 0.3 --updates-per-step 4 > $out/train_fd_u4.log 2>&1
$T --expert-prob 0.3 --batch-episodes 512 --updates-per-step 2 > $out/train_fd_b512_u2.log 2>&1
tail -3 $out/gputests.log
