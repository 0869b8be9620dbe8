cd /root/repo
export NEPMI_JIT=2
(timeout 600 python -m pytest tests/test_jit_shapes.py -m gpu -q -x) 2>&1 | tail -2
timeout 200 python bench.py --no-cpu-baseline --no-extras --workload carbon2024 --reps 10 10 10 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:40], 'ms/step %.4f value %.4g' % (d['ms_per_step'], d['value']), {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()}); print(d['config']['kernel_forms']); print(d.get('thermo_last'))"
