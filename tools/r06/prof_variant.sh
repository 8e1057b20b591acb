#!/bin/bash
# kernel stats of the driver-form bench for a build variant: tools/r06/prof_variant.sh NAME [bench args]
v=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $out
if [ $v != default ]; then export KS_LIB=$GRAFT_REPO_ROOT/tools/experiments/build/libkinova_sim_$v.so; fi
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $out/prof_${v}_bench.log 2>&1
head -8 /tmp/prof_$v/p_kernel_stats.csv | cut -d, -f1-8 | cut -c1-200 > $out/prof_${v}_kernel_stats_head.txt
cat $out/prof_${v}_kernel_stats_head.txt
grep '^{' $out/prof_${v}_bench.log | tail -1 | cut -c1-160
