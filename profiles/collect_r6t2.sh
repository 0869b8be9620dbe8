cd /root/repo
(timeout 600 python -m pytest tests/test_tersoff.py tests/test_ref_md_parity.py -m gpu -q -x -k "tersoff or Tersoff or si_" ) 2>&1 | tail -3
for i in 1 2; do
timeout 120 python bench.py --no-cpu-baseline --no-extras --workload si_tersoff --steps 2000 --warmup 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:40], 'ms/step %.5f value %.4g' % (d['ms_per_step'], d['value']), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
