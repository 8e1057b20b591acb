"""Dev tool: gaps between consecutive stepping launches of a traced bench.py run (rocprofv3 --kernel-trace --output-format csv -d DIR).
Usage: step_gaps.py DIR [n_last_steps]"""
import csv, glob, sys, statistics as st
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
steps = [r for r in rows if "k_env_step" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 61
steps = steps[-n - 1:-1]
per = [int(b["Start_Timestamp"]) - int(a["Start_Timestamp"]) for a, b in zip(steps[:-1], steps[1:])]
dur = [int(a["End_Timestamp"]) - int(a["Start_Timestamp"]) for a in steps[:-1]]
gap = [p - d for p, d in zip(per, dur)]
print("period us mean %.1f, kernel mean %.1f, gap mean %.1f" % (st.mean(per) / 1e3, st.mean(dur) / 1e3, st.mean(gap) / 1e3))
print("kernel us:", [round(d / 1e3) for d in dur])
print("gap us:   ", [round(g / 1e3) for g in gap])
