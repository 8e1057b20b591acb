#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of k_env_step for build / config variants (experiment)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="bench.py --mode sim --steps 6 --warmup 2 --no-cpu-baseline"
run() { # name, env...
  name=$1; shift
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf gpurun_out/ws_$name; mkdir -p gpurun_out/ws_$name
    env "$@" true
    ( export "$@" DUMMY=1; rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/ws_$name -- python3 $ARGS > gpurun_out/ws_$name.log 2>&1 )
    python3 - <<PY
import csv, glob
tot=n=0
for f in glob.glob('gpurun_out/ws_$name/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_env_step' in r['Kernel_Name']: tot+=float(r['Counter_Value']); n+=1
print('$name', '$c', 'per launch KiB', tot/max(n,1), 'launches', n)
PY
  done
}
run default DUMMY2=1
run nopairmem KS_PAIR_MEMORY=0
run outline KS_LIB=$GRAFT_REPO_ROOT/kinovagrasping_amd/libkinova_sim_outl.so
run norays KS_RAYS_IN_STEP=0
