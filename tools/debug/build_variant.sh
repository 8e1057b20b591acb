#!/bin/bash
# bash tools/debug/build_variant.sh NAME -DFLAG ...  -> kinovagrasping_amd/libkinova_sim_NAME.so (for tools/debug/ab.sh style A/B runs on one box)
name=$1; shift
cd $(dirname $0)/../../kinovagrasping_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o ../libkinova_sim_$name.so ks_api.hip ks_rollout.hip ks_mlp.hip ks_xchg.hip
