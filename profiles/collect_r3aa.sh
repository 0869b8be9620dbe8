#!/bin/bash
# r3aa: force assembly with the table rows of the brick's own atoms in LDS (-DNEPMI_FW_OWN=1) against the product
set -x
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
cp gpumd_amd/lib/libnepmi.so /tmp/libnepmi_prod.so
for v in prod own prod own; do
  if [ $v = prod ]; then cp /tmp/libnepmi_prod.so gpumd_amd/lib/libnepmi.so; else cp gpumd_amd/lib/variants/libnepmi_$v.so gpumd_amd/lib/libnepmi.so; fi
  for w in pbte pbte_ortho; do
    timeout 200 python bench.py --no-cpu-baseline --no-extras --workload $w --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $w', '%.4f'%d['ms_per_step'], {k:round(v['avg_ms'],4) for k,v in d['kernels'].items() if isinstance(v,dict) and k in ('radial_descriptor','force_assemble','angular_partial_force')})"
  done
done
for v in prod own; do
  if [ $v = prod ]; then cp /tmp/libnepmi_prod.so gpumd_amd/lib/libnepmi.so; else cp gpumd_amd/lib/variants/libnepmi_$v.so gpumd_amd/lib/libnepmi.so; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --workload carbon --reps 10 10 10 --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v carbon', '%.4f'%d['ms_per_step'], {k:round(v['avg_ms'],4) for k,v in d['kernels'].items() if isinstance(v,dict) and k in ('radial_descriptor','force_assemble','angular_partial_force')})"
done
cp gpumd_amd/lib/variants/libnepmi_own.so gpumd_amd/lib/libnepmi.so
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not full_size" 2>&1 | grep -E "passed|failed" | tail -2
cp /tmp/libnepmi_prod.so gpumd_amd/lib/libnepmi.so
