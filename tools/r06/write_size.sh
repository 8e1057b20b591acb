#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of k_rollout per env-step for build variants (one rocprofv3 --pmc pass each): where does the counter traffic come from?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then unset KS_LIB; else export KS_LIB=$GRAFT_REPO_ROOT/tools/experiments/build/libkinova_sim_$v.so; fi
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/ws_$v; timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/ws_$v -- python3 bench.py --rollout free --chunk 10 --steps 20 --warmup 10 --repeats 1 --no-cpu-baseline --pretrain-updates 150 --steady-steps 0 > /tmp/ws_$v.log 2>&1
    python3 - $v $c <<'PY'
import csv, glob, sys
v, c = sys.argv[1], sys.argv[2]
rows = [r for f in glob.glob(f'/tmp/ws_{v}/*/*counter_collection.csv') for r in csv.DictReader(open(f)) if 'k_rollout' in r['Kernel_Name'] and r['Counter_Name'] == c]
ids = sorted({int(r['Dispatch_Id']) for r in rows})[-2:]
val = sum(float(r['Counter_Value']) for r in rows if int(r['Dispatch_Id']) in ids) / max(1, len(ids)) / 10
print(f"{v:10s} {c}: {val * 1024 / 1e6:8.1f} MB per env-step (k_rollout, last 2 launches of 10)")
PY
  done
done
