"""Dev tool: timeline of one steady-state env-step from a rocprofv3 kernel trace csv (argv[1])."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last but 3rd k_env_step launch and print everything until the next one
idx = [i for i, r in enumerate(rows) if "k_env_step" in r["Kernel_Name"]]
a, b = idx[-4], idx[-3]
t0 = int(rows[a]["Start_Timestamp"])
# include kernels that started up to 0.3 ms before (g_pre)
lo = a
while lo > 0 and int(rows[lo - 1]["Start_Timestamp"]) > t0 - 300000:
    lo -= 1
print("step window %.3f ms" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e6))
busy = {}
for r in rows[lo:b]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"]
    short = name.split("(")[0][-60:]
    q = r.get("Queue_Id", "?")
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f} us  q{q}  {short}")
