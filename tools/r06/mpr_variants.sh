#!/bin/bash
# round 6: long-horizon parity THROUGH ks_step (warm pair memory) and the rate of the penetration query's variants
out=gpurun_out/r06; mkdir -p $out
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then unset KS_LIB; else export KS_LIB=$PWD/tools/experiments/build/libkinova_sim_$v.so; fi
  echo "== $v" >> $out/mpr_variants.txt
  python -m tests.studies.long_horizon_envstep 14 2>&1 | grep -v amdgpu.ids >> $out/mpr_variants.txt
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | cut -c1-160 >> $out/mpr_variants.txt
  python bench.py --mode sim --no-cpu-baseline 2>/dev/null | grep '^{' | cut -c1-160 >> $out/mpr_variants.txt
done
cat $out/mpr_variants.txt
