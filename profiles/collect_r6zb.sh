# round 6, final tree: GPU tier, the driver's bench command, small sizes (adaptive looks at the flags)
cd /root/repo
(timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/pytest_r6zb.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_r6zb.log | head -5
S=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6zb_bench.json 2> gpurun_out/r6zb_bench.err; echo "bench wall $(( $(date +%s) - S )) s"
python -c "
import json; d=json.loads(open('gpurun_out/r6zb_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value']); print({k:round(v['ms_per_step'],4) for k,v in d['like_for_like'].items() if isinstance(v,dict)})
for k in ('config2_si_tersoff','config4_model_unep_1m','config5_model_carbon_1m'): print(k, d['extra_measurements'][k]['ms_per_step'], d['extra_measurements'][k]['value'])"
for wl in "--workload si_tersoff --steps 2000 --warmup 200" "--reps 4 4 4 --steps 400 --warmup 40" "--reps 8 8 8 --steps 200 --warmup 20" "--reps 10 10 10 --steps 200 --warmup 20"; do
  timeout 120 python bench.py --no-cpu-baseline --no-extras $wl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:40], 'ms/step %.5f value %.4g' % (d['ms_per_step'], d['value']))"
done
