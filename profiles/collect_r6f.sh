cd /root/repo
for wm in 0 6656; do
  export NEPMI_WIN_MAX_ATOMS=$wm
  [ $wm = 0 ] && unset NEPMI_WIN_MAX_ATOMS
  NEPMI_JIT=1 NEPMI_FORCE_COVER=1 timeout 200 python bench.py --no-cpu-baseline --no-extras --workload carbon2024 --reps 10 10 10 --steps 5 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cap $wm cover', 'ms/step %.4f' % d['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()}); print(d['config']['kernel_forms'][:230]); print(d.get('thermo_last')[:3])"
  NEPMI_JIT=0 timeout 300 python bench.py --no-cpu-baseline --no-extras --workload carbon2024 --reps 10 10 10 --steps 3 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cap $wm generic', 'ms/step %.4f' % d['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()}); print(d['config']['kernel_forms'][:230]); print(d.get('thermo_last')[:3])"
done
