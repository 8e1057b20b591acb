#!/bin/bash
# Round 5 experiment: what does a private-memory word stored before / loaded after the out-of-line `collision` cost?  
# Builds of the standard library with KS_SCRATCH_PROBE = 0 / 64 / 128 / 256 extra words per lane and substep (ks_core.h: mj_forward_step), timed on the
# sim-only bench (k_env_step) and the default training bench (k_rollout).  Build here (hipcc cross-compiles), run on the GPU box:
#   tools/experiments/scratch_probe.sh build      (in the authoring container)
#   gpurun -- bash tools/experiments/scratch_probe.sh run > profiles/r05_scratch_probe.txt
cd "$(dirname "$0")/../.."
B=tools/experiments/build
if [ "$1" = build ]; then
  for n in 64 128 256; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DKS_SCRATCH_PROBE=$n -o $B/libkinova_sim_probe$n.so kinovagrasping_amd/csrc/ks_api.hip kinovagrasping_amd/csrc/ks_rollout.hip kinovagrasping_amd/csrc/ks_mlp.hip kinovagrasping_amd/csrc/ks_xchg.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "Function Name: .*k_rolloutILi16ELi16" | grep -E "Scratch" | sed "s/^.*remark:/probe $n k_rollout:/"
  done
else
  echo "KS_SCRATCH_PROBE words per lane and substep (stored before / loaded after collision) -> env-steps/s, ms per env-step (bench.py, 4096 envs, 1 x MI355X)"
  for n in 0 64 128 256; do
    lib=kinovagrasping_amd/libkinova_sim.so; [ $n != 0 ] && lib=$B/libkinova_sim_probe$n.so
    for mode in "--mode sim" ""; do
      KS_LIB=$PWD/$lib python3 bench.py $mode --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('probe %3d  %-10s %.3f M env-steps/s  %.4f ms per env-step  (kernel %.4f ms)' % ($n, '$mode' or 'training', d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms']))"
    done
  done
fi
