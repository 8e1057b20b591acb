#!/bin/bash
# Dev tool: rocprofv3 PMC passes over the bench (separate passes; --kernel-trace only, as the pool requires).
# usage: bash tools/pmc_run.sh [sim|ddpg|free|mg] [outdir]     - counters of the LAST 40 k_env_step launches of every pass are averaged
#   sim : bench.py --mode sim (config 2 at 4096 envs)          ddpg: bench.py --eager (config 3, learner launched op by op) after 600 untimed pre-training updates
mode=${1:-sim}
dir=${2:-gpurun_out/pmc_$mode}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $dir; mkdir -p $dir
KERNEL=k_env_step; FILTER="--kernel-include-regex k_env_step"
if [ $mode = free ]; then
  # the free-running rollout kernel: the learner replays HIP graphs beside it, the kernel filter segfaults with those -> counters on every
  # dispatch (the collector segfaults at 600 pre-training updates, 150 works), k_rollout launches of 10 env-steps each (divide the per-launch figures by 10)
  ARGS="bench.py --rollout free --chunk 10 --steps 20 --warmup 10 --no-cpu-baseline --pretrain-updates ${PMC_PRETRAIN:-150} --steady-steps 0"; KERNEL=k_rollout; FILTER=""
elif [ $mode = sim ]; then ARGS="bench.py --mode sim --steps 40 --warmup 4 --no-cpu-baseline"
elif [ $mode = mg ]; then ARGS="bench.py --mode sim --shape ${PMC_SHAPE:-BowlS} --steps 40 --warmup 4 --no-cpu-baseline"      # libkinova_sim_mg.so: a multi-geom object (round 5)
else ARGS="bench.py --rollout lockstep --eager --steps 40 --warmup 5 --no-cpu-baseline --pretrain-updates 600 --steady-steps 0"; fi   # (--eager: counter collection + the kernel filter segfaults rocprofv3 when the learner runs from HIP graphs)
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" \
           "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL" \
           "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace $FILTER --pmc $set --output-format csv -d $dir/p$i -- python3 $ARGS > $dir/p$i.log 2>&1
done
PMC_DIR=$dir PMC_KERNEL=$KERNEL python3 - <<'PY'
import csv, glob, collections, os
d = os.environ["PMC_DIR"]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + '/p*/*/*counter_collection.csv'):
    rows = [r for r in csv.DictReader(open(f)) if os.environ["PMC_KERNEL"] in r['Kernel_Name']]
    ids = sorted({int(r['Dispatch_Id']) for r in rows})[-(2 if os.environ["PMC_KERNEL"] == "k_rollout" else 40):]
    keep = set(ids)
    for r in rows:
        if int(r['Dispatch_Id']) in keep:
            a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
with open(d + '/summary.txt', 'w') as out:
    for k in sorted(agg):
        line = f"{k:28s} per-launch avg {agg[k][0]/agg[k][1]:16.1f}  launches {agg[k][1]}"
        print(line); out.write(line + "\n")
PY
rm -rf $dir/p*/
