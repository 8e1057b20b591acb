# Dev tool: average duration of k_rays / k_obs / k_env_step over the sim-only bench (rocprofv3 kernel stats)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sim -o sim -- python3 $GRAFT_REPO_ROOT/bench.py --mode sim --no-cpu-baseline > /tmp/sim.log 2>&1
python3 - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/prof_sim/sim_kernel_stats.csv')):
    if any(k in r['Name'] for k in ('k_rays', 'k_obs', 'k_env_step', 'k_reset')):
        print(r['Name'][:60], r['Calls'], 'avg us', float(r['AverageNs']) / 1e3, 'min', float(r['MinNs']) / 1e3, 'max', float(r['MaxNs']) / 1e3)
PY
