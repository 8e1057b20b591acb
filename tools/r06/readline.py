import json
d=json.loads([l for l in open("gpurun_out/r06/driver_form_full.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["timed_windows_ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
print(d["roofline"]["issue_bound"])
print(d["cpu_baseline"])
print({k: d["config"][k] for k in list(d["config"])[:8]})
