#!/bin/bash
# r3x: split-lane window kernels as kLanes workgroups of 256 threads per brick (NEPMI_WIN_PARTS=1, default) against one of 256 kLanes
set -x
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
for parts in 1 0; do
  for r in 4 5 6; do
    NEPMI_WIN_PARTS=$parts timeout 120 python bench.py --no-cpu-baseline --no-extras --reps $r $r $r --steps 400 --warmup 40 > gpurun_out/r3x_r${r}_parts$parts.json 2>/dev/null
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3x_r*_parts*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["config"].get("atoms"), "ms/step %.4f"%d["ms_per_step"], "%.3e"%d["value"], {k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if isinstance(v,dict) and "avg_ms" in v})
    except Exception as e: print(f,"ERR",e)
PY
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_model_variants.py -m gpu -q -x -k "not full_size" > gpurun_out/r3x_pytest.log 2>&1; tail -2 gpurun_out/r3x_pytest.log
