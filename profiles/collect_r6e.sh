cd /root/repo
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "UNEP") 2>&1 | tail -2
for i in 1 2; do
timeout 200 python bench.py --no-cpu-baseline --no-extras --workload unep --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:40], 'ms/step %.4f value %.4g' % (d['ms_per_step'], d['value']), {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()}); print(d.get('thermo_last')[:3])"
done
