#!/usr/bin/env python3
"""profiles/<tag>_pmc_k_env_step_summary.txt (tools/pmc_run.sh) -> profiles/<round>_pmc.json, the file bench.py reads for
`roofline.traffic` and the issue-bound figures.  Usage: tools/pmc_to_json.py profiles/r04_a_pmc_sim_summary.txt profiles/r04_pmc_sim.json sim|ddpg|free"""
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "sim"
per = {}
for line in open(src):
    m = re.match(r"(\S+)\s+per-launch avg\s+([0-9.eE+-]+)", line)
    if m:
        per[m.group(1)] = float(m.group(2))
kib = 1024.0
if workload == "free":          # k_rollout launches advance 10 env-steps each: normalise to ONE env-step so the figures compare with k_env_step's
    per = {k: (v if k == "SQ_WAVES" else v / 10.0) for k, v in per.items()}
out = {
    "kernel": "k_rollout" if workload == "free" else "k_env_step",
    "workload": ("bench.py --mode sim, 4096 envs, CubeS" if workload == "sim" else
                 "bench.py --mode sim --shape BowlS, 4096 envs: libkinova_sim_mg.so (hull tables in global memory, 9 hulls on the floor, ~15 contacts per env at rest)" if workload == "mg" else
                 "bench.py --rollout free (config 3: DDPG training, 4096 envs, free-running rollout kernel, learner HIP graphs; counters on every dispatch, "
                 "dispatches serialised by the collector so k_rollout runs alone; per-launch figures divided by the 10 env-steps of a launch) after 150 pre-training updates from the committed bench policy (bench.py's default is 300: with 300 the collector hangs - round 6, a run killed at the 30-minute limit - and with 600 it segfaults; the regime is pinned by the committed policy, not by their number)" if workload == "free" else
                 "bench.py --eager (config 3: DDPG training, 4096 envs, learner launched op by op - counter collection with the kernel filter segfaults when the learner runs from HIP graphs) after 600 pre-training updates") +
                " (tools/pmc_run.sh: rocprofv3 --kernel-trace --pmc, one counter set per pass, last 40 launches of each pass; free: last 2 launches = 20 env-steps)",
    "source": src,
    "per_launch": per,
    # FETCH_SIZE / WRITE_SIZE are reported in KiB.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the
    # bytes of WIDE (16 B / lane) coalesced reads; this kernel's global reads are dword-per-lane gathers (state, slot list,
    # pair memory, cube-map tables, ray-mesh nodes) for which the width is uncalibrated, so the read side is given as
    # measured with the doubled value as an upper bound.  WRITE_SIZE is exact for streaming stores.
    "fetch_bytes_per_launch": per.get("FETCH_SIZE", 0) * kib,
    "fetch_bytes_per_launch_upper_bound": 2 * per.get("FETCH_SIZE", 0) * kib,
    "write_bytes_per_launch": per.get("WRITE_SIZE", 0) * kib,
    "hbm_bytes_per_launch": (per.get("FETCH_SIZE", 0) + per.get("WRITE_SIZE", 0)) * kib,
    "note": ("per env-step of k_rollout: fetch = the actor's weights for every workgroup and env-step (L2 hits) + env state + pair memory + the rays' mesh nodes + "
             "1/10 of the per-launch staging of the hull / model tables; write = state + body-pose snapshot + rays + observation (x3: output, next policy input, "
             "terminal) + replay row + pair memory, stored as 4-byte columns of [field][env] arrays (~12 MB of output stores per env-step) + write-backs of the "
             "private-memory frame (round 6, final: 360 B per lane under the kernel's register budget - spilled values of the inlined solver and of the fp64 penetration query; the two-lane query keeps its "
             "portal in LDS, the one-lane form's 96-byte private array is gone; evicted lines, not a count of the stores - what a frame costs was measured in profiles/r05_scratch_probe.txt; with the "
             "query out of line its 34 callee-saved registers alone made this 144 MB: profiles/r06_mpr_variants.txt), DESIGN section 5.  The counters sit on the L2's memory side: Infinity-Cache hits are included, "
             "so this is an upper bound on HBM traffic.") if workload == "free" else
            "fetch = per-workgroup staging of the hull / model tables (256 workgroups x ~50 KB, L2 / MALL hits count) + env state + "
            "pair memory + the in-step rays' mesh nodes; write = state + snapshot + rays + pair memory (4.9 MB) + write-through of the "
            "private-memory (stack) stores of the out-of-line stages (~4 MB: the pair memory words and an out-of-line fallback's result).  The counters sit on the L2's memory side: Infinity-Cache hits "
            "are included, so this is an upper bound on HBM traffic.",
}
if "SQ_THREAD_CYCLES_VALU" in per and "SQ_ACTIVE_INST_VALU" in per:
    out["valu_lane_efficiency"] = per["SQ_THREAD_CYCLES_VALU"] / (64.0 * per["SQ_ACTIVE_INST_VALU"])
if "SQ_LDS_BANK_CONFLICT" in per and "SQ_LDS_IDX_ACTIVE" in per:
    out["lds_bank_conflict_frac"] = per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"]
if "TCC_HIT_sum" in per:
    out["l2_hit_rate"] = per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
json.dump(out, open(dst, "w"), indent=1)
print({k: v for k, v in out.items() if k not in ("per_launch", "note")})
