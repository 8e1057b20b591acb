#!/bin/bash
# free-running rollout: learner with the one-wave kernels (default) against a build whose split kernels are capped at 128 registers
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
python3 tools/debug/async_parts.py 10
KS_LIB=$PWD/kinovagrasping_amd/libkinova_sim_SPLIT128.so KS_ASYNC_MLP_SPLIT=4 python3 tools/debug/async_parts.py 10
done
