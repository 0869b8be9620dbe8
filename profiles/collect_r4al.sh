# round 4: lanes per atom of the window kernels between 400 and 1000 bricks (PbTe): where does one lane per atom start to win?
cd /root/repo
for reps in "8 8 8" "8 8 10" "8 10 10" "10 10 10"; do
  for L in 1 2; do
    NEPMI_BENCH_LANES=$L timeout 120 python bench.py --no-cpu-baseline --no-extras --reps $reps --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('reps $reps lanes $L atoms', d['config']['atoms_total'], 'ms/step %.4f value %.4g'%(d['ms_per_step'], d['value']), {k[:10]:round(v['avg_ms'],4) for k,v in d['kernels'].items() if k!='list_rebuild'}, d['config']['kernel_forms'].split('bricks=')[1])"
  done
done 2>&1 | tee gpurun_out/r4al_lanes_sweep.txt
