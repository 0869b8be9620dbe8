import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), d["roofline"].get("second"))
print({k: (round(v["avg_ms"], 4), round(v.get("frac", 0), 3), v["timed_in"][:5]) for k, v in d["kernels"].items()})
for k, v in d.get("extra_measurements", {}).items():
    print(k, v.get("ms_per_step"), v.get("value"), v.get("error"), v.get("kernels_avg_ms"))
print(d.get("gpu_state"))
cb = d.get("cpu_baseline", {})
print(cb.get("value"), cb.get("reference_gpu_same_box", {}).get("value"))
