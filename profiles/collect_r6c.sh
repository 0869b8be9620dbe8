cd /root/repo
export NEPMI_JIT=2 NEPMI_DEBUG_REBUILD=1
NEPMI_WIN_MAX_ATOMS=5000 timeout 200 python tools/c2024_probe.py 24 gpurun_out/c24_gather.npz 2>&1 | grep -v amdgpu.ids | cut -c1-300
NEPMI_WIN_MAX_ATOMS=6656 timeout 200 python tools/c2024_probe.py 24 gpurun_out/c24_window.npz 2>&1 | grep -v amdgpu.ids | cut -c1-300
python tools/c2024_probe.py --compare gpurun_out/c24_gather.npz gpurun_out/c24_window.npz
unset NEPMI_DEBUG_REBUILD
for wm in 5000 6656; do
  NEPMI_WIN_MAX_ATOMS=$wm timeout 200 python bench.py --no-cpu-baseline --no-extras --workload carbon2024 --reps 10 10 10 --steps 10 --warmup 3 2>gpurun_out/r6c_err_$wm.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('win_max $wm', d['config']['workload'][:40], 'ms/step %.4f value %.4g' % (d['ms_per_step'], d['value']), {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()}); print(d['config']['kernel_forms']); print(d.get('thermo_last'))"
  tail -2 gpurun_out/r6c_err_$wm.txt | cut -c1-300
done
