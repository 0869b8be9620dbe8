#!/bin/bash
# r3z1: UNEP force assembly (Fp-row form) with 16-byte LDS reads of the coefficient blocks (-DNEPMI_CT_VEC_FORCE=1) against the product
set -x
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
cp gpumd_amd/lib/libnepmi.so /tmp/libnepmi_prod.so
for v in prod ctvec; do
  if [ $v = prod ]; then cp /tmp/libnepmi_prod.so gpumd_amd/lib/libnepmi.so; else cp gpumd_amd/lib/variants/libnepmi_$v.so gpumd_amd/lib/libnepmi.so; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --workload unep --steps 20 --warmup 5 > gpurun_out/r3z1_unep_$v.json 2>/dev/null
done
cp /tmp/libnepmi_prod.so gpumd_amd/lib/libnepmi.so
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3z1_unep_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "ms/step %.4f"%d["ms_per_step"], {k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if isinstance(v,dict) and "avg_ms" in v}, d["config"].get("kernel_forms"))
PY
