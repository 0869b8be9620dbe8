# round 4: Verlet lists of a rebuild from LDS windows (BuildListsWinBody) against the gather form (BuildListsBody)
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
cp gpumd_amd/lib/libnepmi.so /tmp/libnepmi_product.so
for wl in "--steps 100 --warmup 10" "--workload carbon --reps 10 10 10 --steps 60 --warmup 10" "--workload unep --steps 60 --warmup 10"; do
for v in gpumd_amd/lib/variants/libnepmi_*.so; do
  name=$(basename $v .so); name=${name#libnepmi_}
  cp $v gpumd_amd/lib/libnepmi.so
  python bench.py --no-cpu-baseline --no-extras $wl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s'%'$name', '$wl', 'ms/step %.4f'%d['ms_per_step'], 'rebuilds', d['config']['rebuilds_in_timed_region'], 'list_rebuild ms', round(d['kernels'].get('list_rebuild',{}).get('avg_ms',0),3))"
done; done 2>&1 | tee gpurun_out/r4an_ab_build_win.txt
cp /tmp/libnepmi_product.so gpumd_amd/lib/libnepmi.so
