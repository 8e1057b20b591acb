"""Dev tool: kernels of the learner's update replayed alone (the last launches on the learner's stream of a bench.py run traced
with `rocprofv3 --kernel-trace --output-format csv -d DIR`): duration of and gap before each.  Usage: update_timeline.py DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
KEYS = ("k_env_step", "k_mlp3_wave", "k_mlp3_bwd_wave", "k_wgrad_wave", "k_wgrad_reduce", "k_mlp3<", "k_rank", "k_commit", "k_store", "k_sample",
        "k_adam", "k_soft", "k_critic_grad", "k_update_prologue", "k_advance", "k_xchg")
def short(n):
    for k in KEYS:
        if k in n:
            return k
    return n[:30]
main = [r for r in rows if "k_env_step" in r["Kernel_Name"]][-1]["Stream_Id"]
side = [r for r in rows if r["Stream_Id"] != main]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 46
prev, tk, tg = None, 0.0, 0.0
for r in side[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%-20s dur %7.1f  gap %6.1f" % (short(r["Kernel_Name"]), (e - s) / 1e3, gap))
    tk += (e - s) / 1e3; tg += max(gap, 0.0); prev = e
print("kernels %.0f us, gaps %.0f us" % (tk, tg))
