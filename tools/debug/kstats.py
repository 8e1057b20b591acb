"""Dev tool: first rows of a rocprofv3 kernel-stats csv found under a directory.  Usage: kstats.py <dir> [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  min {float(r["MinNs"]) / 1e3:8.1f}  max {float(r["MaxNs"]) / 1e3:8.1f}')
