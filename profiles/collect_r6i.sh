# round 6i: the fused angular kernel for many-type models (type-sorted work order, window of four types in LDS) -- parity, then A/B on UNEP-v1
cd /root/repo
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_model_variants.py -m gpu -q -x) > gpurun_out/pytest_r6i.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_r6i.log | head
for f in 1 0; do
NEPMI_BENCH_ANGFUSED=$f timeout 300 python bench.py --no-cpu-baseline --no-extras --workload unep --reps 16 16 16 --steps 20 --warmup 5 > gpurun_out/bench_r6i_u_f$f.json 2> gpurun_out/bench_r6i_u_f$f.err
python - gpurun_out/bench_r6i_u_f$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("unep ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
    print(d["config"]["kernel_forms"])
except Exception as e:
    print(sys.argv[1], "ERR", e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done
