#!/bin/bash
# round 6: parity (query bit-equality against a second build, long horizon through ks_step) and rates of the default build against build variants
# usage: VARIANTS="a default a default" REF=a tools/r06/ab_generic.sh tag
tag=${1:-ab}; mkdir -p gpurun_out/r06s2; o=gpurun_out/r06s2/$tag.txt; : > $o
B=$PWD/tools/experiments/build
python tools/r06/ab_bits.py run /tmp/a.npz 8 2>/dev/null; KS_LIB=$B/libkinova_sim_${REF}.so python tools/r06/ab_bits.py run /tmp/b.npz 8 2>/dev/null
echo "default vs $REF: $(python tools/r06/ab_bits.py cmp /tmp/a.npz /tmp/b.npz | tail -1)" >> $o
VARIANTS="$VARIANTS" bash tools/r06/rate_variants.sh >> $o 2>&1
cat $o
