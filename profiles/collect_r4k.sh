#!/bin/bash
# r4k: whole GPU tier + the bench command the driver runs (extras included)
set -x
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4k
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/${T}_pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/${T}_pytest_gpu.log
grep -E "full-size parity" gpurun_out/${T}_pytest_gpu.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver.json 2> gpurun_out/${T}_bench_driver.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r4k_bench_driver.json').read().strip().splitlines()[-1])
print("ms/step %.4f value %.4g frac %.3f"%(d["ms_per_step"], d["value"], d["roofline"]["frac"]), {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()})
for k,v in d.get("extra_measurements",{}).items():
    print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("ms_per_step","value","error","rebuilds_in_timed_region")})
print(d["cpu_baseline"]["value"], d["cpu_baseline"].get("reference_gpu_same_box",{}).get("value"))
PY
