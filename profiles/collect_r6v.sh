cd /root/repo
(timeout 1500 python -m pytest tests -m gpu -q -x) > gpurun_out/pytest_r6v.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_r6v.log | head -5
S=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6v_bench.json 2> gpurun_out/r6v_bench.err; echo "bench wall $(( $(date +%s) - S )) s"
python -c "
import json; d=json.loads(open('gpurun_out/r6v_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value']); print({k:round(v['ms_per_step'],4) for k,v in d['like_for_like'].items() if isinstance(v,dict)})
for k in ('config2_si_tersoff','config4_model_unep_1m','config5_model_carbon_1m'): print(k, d['extra_measurements'][k]['ms_per_step'])"
