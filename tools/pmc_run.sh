#!/bin/bash
# Dev tool: rocprofv3 PMC passes over the sim-only bench (separate passes; --kernel-trace only, as the pool requires)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="bench.py --mode sim --steps 6 --warmup 2 --no-cpu-baseline"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" \
           "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL" \
           "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_r2/p$i -- python3 $ARGS > gpurun_out/pmc_r2/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/pmc_r2/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_env_step' in r['Kernel_Name']:
            a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
with open('gpurun_out/pmc_r2/summary.txt', 'w') as out:
    for k in sorted(agg):
        line = f"{k:28s} per-launch avg {agg[k][0]/agg[k][1]:16.1f}  launches {agg[k][1]}"
        print(line); out.write(line + "\n")
PY
