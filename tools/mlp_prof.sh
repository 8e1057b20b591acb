# Dev tool: kernel durations of tools/mlp_bench.py (rocprofv3 kernel stats)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mlp -o mlp -- python3 $GRAFT_REPO_ROOT/tools/mlp_bench.py > /tmp/mlp.log 2>&1
python3 - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/prof_mlp/mlp_kernel_stats.csv')):
    if 'k_mlp3' in r['Name']:
        print(r['Name'][:75], r['Calls'], 'avg us', round(float(r['AverageNs']) / 1e3, 1), 'min', float(r['MinNs']) / 1e3, 'max', float(r['MaxNs']) / 1e3)
PY
