"""Dev tool: every kernel between the ends of the last stepping launches of a bench.py run traced with
`rocprofv3 --kernel-trace --output-format csv -d DIR`: start / end relative to the previous stepping kernel's end, stream and queue.
Usage: step_timeline.py DIR [steps back from the end]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
steps = [r for r in rows if "k_env_step" in r["Kernel_Name"]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = steps[-back - 1], steps[-back + 1] if back > 1 else steps[-1]
t0 = int(a["End_Timestamp"])
lo, hi = int(a["Start_Timestamp"]), int(b["End_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e < lo or s > hi:
        continue
    name = r["Kernel_Name"]
    name = name.split("(anonymous namespace)::")[-1][:34]
    print("%-34s start %9.1f  end %9.1f  dur %7.1f  stream %s queue %s" % (name, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Stream_Id"), r.get("Queue_Id")))
