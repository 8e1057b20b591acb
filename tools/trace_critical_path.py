"""Dev tool: per-step critical-path marks (us from the start of k_env_step) from a rocprofv3 kernel trace csv:
end of k_env_step, end of the simulator's last kernel (k_obs after the reset), start/end of the replay-write graph
(k_store_transition .. k_advance_ring), end of the learner body (last kernel on the other queue), first kernel of
the next action-selection graph, start of the next k_env_step."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
S = lambda r: int(r["Start_Timestamp"])
E = lambda r: int(r["End_Timestamp"])
idx = [i for i, r in enumerate(rows) if "k_env_step" in r["Kernel_Name"]]
print("step  env_end  sim_end  post_start post_end  body_end  pre_start  next_env   head_start head_end")
for k in range(len(idx) - 24, len(idx) - 1):
    a, b = idx[k], idx[k + 1]
    t0 = S(rows[a]); q = rows[a]["Queue_Id"]
    seg = rows[a:b]
    mine = [r for r in seg if r["Queue_Id"] == q]
    oth = [r for r in seg if r["Queue_Id"] != q]
    u = lambda t: (t - t0) / 1e3
    obs = [r for r in mine if "k_obs" in r["Kernel_Name"]]
    st = [r for r in mine if "k_store_transition" in r["Kernel_Name"]]
    adv = [r for r in mine if "k_advance_ring" in r["Kernel_Name"]]
    after = [r for r in mine if adv and S(r) > E(adv[0])]
    samp = [r for r in oth if "k_sample_windows" in r["Kernel_Name"]]
    print(f"{k:4d} {u(E(rows[a])):8.0f} {u(E(obs[-1])) if obs else -1:8.0f} {u(S(st[0])) if st else -1:10.0f} {u(E(adv[0])) if adv else -1:8.0f} "
          f"{u(max(E(r) for r in oth)) if oth else -1:9.0f} {u(S(after[0])) if after else -1:10.0f} {u(S(rows[b])):9.0f}"
          f"   {u(min(S(r) for r in oth)) if oth else -1:8.0f} {u(E(samp[0])) if samp else -1:8.0f}")
