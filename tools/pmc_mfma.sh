#!/bin/bash
# Dev tool: matrix-pipe counters of the learner's kernels (bench.py in training mode), one rocprofv3 --pmc pass (--kernel-trace only,
# as the pool requires).  Output: gpurun_out/pmc_mfma/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_mfma
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > gpurun_out/pmc_mfma/avail.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_mfma/p1 -- python3 bench.py --steps 12 --warmup 40 --no-cpu-baseline --steady-updates 0 > gpurun_out/pmc_mfma/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc_mfma/p2 -- python3 bench.py --steps 12 --warmup 40 --no-cpu-baseline --steady-updates 0 > gpurun_out/pmc_mfma/p2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/pmc_mfma/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(anonymous namespace)::')[-1].split('(')[0][:40]
        a = agg[n][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for f in glob.glob('gpurun_out/pmc_mfma/p1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(anonymous namespace)::')[-1].split('(')[0][:40]
        d = dur[n]; d[0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); d[1] += 1
with open('gpurun_out/pmc_mfma/summary.txt', 'w') as out:
    for n in sorted(agg):
        if not any(k in n for k in ('k_mlp3', 'k_wgrad_wave', 'k_env_step')):
            continue
        line = f"{n:42s} launches {dur[n][1]:5d} avg {dur[n][0] / max(dur[n][1], 1) / 1e3:8.1f} us (serialised by the counter pass) | " + "  ".join(
            f"{c} {v[0] / v[1]:.0f}" for c, v in sorted(agg[n].items()))
        print(line); out.write(line + "\n")
PY
cat gpurun_out/pmc_mfma/avail.txt | tr '\n' ' '
