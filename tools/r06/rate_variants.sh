#!/bin/bash
# rates only (driver form + sim-only) of build variants: VARIANTS="a b" tools/r06/rate_variants.sh
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then unset KS_LIB; else export KS_LIB=$PWD/tools/experiments/build/libkinova_sim_$v.so; fi
  a=$(python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3), d['ms_per_step'])")
  b=$(python bench.py --mode sim --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3), d['ms_per_step'])")
  echo "$v: training (driver form) $a   sim-only $b"
done
