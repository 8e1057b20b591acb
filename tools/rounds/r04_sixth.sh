#!/bin/bash
# round 4: target-rate sweep of DDPGfD (30 % expert mix) around tau = 0.003 per update, 12000 env-steps each; the checkpoints are kept
out=$GRAFT_REPO_ROOT/gpurun_out/r04g
mkdir -p $out
cd $GRAFT_REPO_ROOT
T="python examples/train_ddpgfd.py --envs 4096 --steps 12000 --free-running --expert-prob 0.3 --target-every 1"
$T --tau 0.003 --save $out/fd_tau003 > $out/train_fd_tau003.log 2>&1
$T --tau 0.002 --save $out/fd_tau002 > $out/train_fd_tau002.log 2>&1
$T --tau 0.001 --save $out/fd_tau001 > $out/train_fd_tau001.log 2>&1
$T --tau 0.003 --actor-lr 3e-5 --save $out/fd_tau003_alr > $out/train_fd_tau003_alr3e-5.log 2>&1
$T --tau 0.003 --critic-lr 3e-4 --save $out/fd_tau003_clr > $out/train_fd_tau003_clr3e-4.log 2>&1
python examples/train_ddpgfd.py --envs 4096 --steps 12000 --free-running --expert-prob 0 --target-every 1 --tau 0.003 > $out/train_plain_tau003.log 2>&1
for f in $out/train_*.log; do echo "== $f: $(grep -o 'lift success [0-9.]*' $f | awk '{printf "%s ", $3}')"; done
