#!/bin/bash
# Diagnostic / experiment builds of the standard library: tools/r06/build_variant.sh NAME [-DFLAG ...] -> tools/experiments/build/libkinova_sim_NAME.so
# (git-ignored, travels with gpurun; run a tool against it with KS_LIB=$PWD/tools/experiments/build/libkinova_sim_NAME.so)
set -e
name=$1; shift
cd "$(dirname "$0")/../../kinovagrasping_amd/csrc"
mkdir -p ../../tools/experiments/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o ../../tools/experiments/build/libkinova_sim_$name.so ks_api.hip ks_rollout.hip ks_mlp.hip ks_xchg.hip
ls -l ../../tools/experiments/build/libkinova_sim_$name.so
