#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of k_rollout alone, with and without the pair memory
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 1 0; do for c in WRITE_SIZE FETCH_SIZE; do
  d=gpurun_out/rt_$v$c; rm -rf $d; mkdir -p $d
  export KS_PAIR_MEMORY=$v
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 tools/debug/rollout_traffic.py > $d.log 2>&1
  python3 - <<PY
import csv, glob
v=[]
for f in glob.glob('$d/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_rollout' in r['Kernel_Name']: v.append(float(r['Counter_Value']))
print('pair_memory=$v', '$c', 'per env-step MB', sum(v[-6:])/max(1,len(v[-6:]))/10*1024/1e6, 'launches', len(v))
PY
  rm -rf $d
done; done
