#!/bin/bash
# r4h: many-type scatter form (UNEP-v1) + the halo with one message per peer rank
set -x
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4h
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scatter" > gpurun_out/${T}_pytest_scatter.log 2>&1
tail -5 gpurun_out/${T}_pytest_scatter.log
for form in 0 1; do
  NEPMI_BENCH_FORM=$form timeout 200 python bench.py --no-cpu-baseline --no-extras --workload unep --steps 20 --warmup 5 > gpurun_out/${T}_bench_unep_form$form.json 2>/dev/null
done
NEPMI_BENCH_FORM=1 timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10 > gpurun_out/${T}_bench_form1.json 2>/dev/null
for f in gpurun_out/${T}_bench*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()}, d["config"].get("kernel_forms","")[-70:])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
timeout 1500 python -m pytest tests/test_dist.py tests/test_dist_inproc.py -x -q -m gpu > gpurun_out/${T}_pytest_dist.log 2>&1
grep -E "passed|failed" gpurun_out/${T}_pytest_dist.log
timeout 300 python profiles/inproc_weak.py --ranks 2 --steps 100 > gpurun_out/${T}_weak2.json 2>/dev/null; cut -c1-330 gpurun_out/${T}_weak2.json
for g in 0 1; do
  timeout 300 python profiles/inproc_weak.py --strong --ranks 8 --steps 60 --ghosts $g > gpurun_out/${T}_strong8_g${g}.json 2>/dev/null; grep -o '"ms_per_step_ranks[^,]*, "ms_per_step_one[^,]*, "work_inflation[^,]*' gpurun_out/${T}_strong8_g${g}.json
done
