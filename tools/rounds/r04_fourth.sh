#!/bin/bash
# round 4, fourth batch: multi-group persistent rollout (config 5 free-running), target-network rate sweeps
out=$GRAFT_REPO_ROOT/gpurun_out/r04e
mkdir -p $out
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_async.py tests/test_bench_launch.py tests/test_gpu_config5.py -m gpu -q -s > $out/gputests_async.log 2>&1; echo "pytest rc $?" >> $out/gputests_async.log
python bench.py --no-cpu-baseline --config 5 > $out/bench_cfg5_free.log 2>&1
python bench.py --no-cpu-baseline --config 5 --rollout lockstep > $out/bench_cfg5_lock.log 2>&1
python bench.py --no-cpu-baseline > $out/bench_default.log 2>&1
python bench.py --no-cpu-baseline --mode sim > $out/bench_sim.log 2>&1
T="python examples/train_ddpgfd.py --envs 4096 --steps 9000 --free-running"
$T --expert-prob 0 --tau 0.01 --target-every 1 > $out/train_plain_tau01.log 2>&1
$T --expert-prob 0.3 --tau 0.01 --target-every 1 > $out/train_fd_tau01.log 2>&1
$T --expert-prob 0.3 --tau 0.003 --target-every 1 > $out/train_fd_tau003.log 2>&1
$T --expert-prob 0.3 --tau 0.03 --target-every 1 > $out/train_fd_tau03.log 2>&1
$T --expert-prob 0.3 --tau 0.01 --target-every 1 --updates-per-step 2 > $out/train_fd_tau01_u2.log 2>&1
$T --expert-prob 0.3 --tau 0.01 --target-every 1 --expl-noise 0.05 > $out/train_fd_tau01_noise05.log 2>&1
$T --expert-prob 0.3 --tau 0.01 --target-every 1 --actor-lr 3e-5 > $out/train_fd_tau01_alr3e-5.log 2>&1
tail -3 $out/gputests_async.log
