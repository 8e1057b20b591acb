"""Dev tool: per-step summary (k_env_step, k_rays, learner span, step window) from a rocprofv3 kernel trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_env_step" in r["Kernel_Name"]]
print("step  window   k_env_step  k_rays(max)  other-queue busy span  kernels")
for k in range(len(idx) - 32, len(idx) - 1):
    a, b = idx[k], idx[k + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    win = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
    kes = (int(rows[a]["End_Timestamp"]) - t0) / 1e3
    qa = rows[a].get("Queue_Id")
    rays = max([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[a:b] if "k_rays" in r["Kernel_Name"]] + [0])
    oth = [r for r in rows[a:b] if r.get("Queue_Id") != qa]
    span = ((max(int(r["End_Timestamp"]) for r in oth) - min(int(r["Start_Timestamp"]) for r in oth)) / 1e3) if oth else 0
    first = ((min(int(r["Start_Timestamp"]) for r in oth) - t0) / 1e3) if oth else 0
    last = ((max(int(r["End_Timestamp"]) for r in oth) - t0) / 1e3) if oth else 0
    print(f"{k - (len(idx) - 32):3d} {win:8.0f} {kes:10.0f} {rays:10.0f}   learner {first:6.0f} -> {last:6.0f} ({len(oth)} kernels)   total kernels {b - a}")
