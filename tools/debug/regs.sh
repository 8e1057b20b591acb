#!/bin/bash
# register footprint of the fp32 stepping kernel and its out-of-line stages (the learner's kernels are compiled for 3 waves
# per SIMD = 168 registers, so k_env_step has to stay <= 344 for them to run beside it)
cd "$(dirname "$0")/../../kinovagrasping_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o /tmp/ks_api.s ks_api.hip "$@" 2>/dev/null
awk '/^\t\.size\t_Z/{name=$2} /; codeLenInByte/{len=$4} /; NumVgprs:/{v=$3} /; NumAgprs:/{a=$3; if (name !~ /Id/) printf "%-60s code %7d  vgpr %3d agpr %3d\n", substr(name,1,60), len, v, a}' /tmp/ks_api.s
