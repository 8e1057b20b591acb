#!/bin/bash
# phase stamps + per-workgroup loop times of the free-running rollout kernel (diagnostic build libkinova_sim_stamp.so: build_variant.sh stamp -DKS_ROLLOUT_STAMP)
cd $GRAFT_REPO_ROOT
KS_LIB=$PWD/kinovagrasping_amd/libkinova_sim_stamp.so python3 tools/debug/async_parts.py ${1:-10}
