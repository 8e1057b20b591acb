#!/bin/bash
# round 5, third batch: DDPGfD in the reference's schedule, three times as long (9000 env-steps of 32 envs = 963 000 updates), and at 64 envs
out=$GRAFT_REPO_ROOT/gpurun_out/r05_c
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 2400 python3 examples/train_ddpgfd.py --envs 32 --steps 9000 --updates-per-step 107 --tau 0.0005 --target-every 10 --eval-every 600 > $out/train_ref_regime_32_long.log 2>&1
grep -i "eval" $out/train_ref_regime_32_long.log | tail -16
