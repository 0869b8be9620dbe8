# round 5: fused angular kernel on carbon (lane pairs, 5 channels per lane, weight rows read twice) -- parity + same-box A/B
cd /root/repo
T=r5b
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_angular or test_force_parity") > gpurun_out/${T}_pytest.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/${T}_pytest.log | head -20
for v in 1 0 1 0; do
NEPMI_BENCH_ANGFUSED=$v python bench.py --no-cpu-baseline --no-extras --workload carbon --reps 10 10 10 --steps 30 --warmup 5 > gpurun_out/${T}_bench_c_f$v.json 2> gpurun_out/${T}_bench_c_f$v.err
python - gpurun_out/${T}_bench_c_f$v.json $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("carbon fused=%s ms/step %.4f value %.4g"%(sys.argv[2], d["ms_per_step"], d["value"]), {k[:12]:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
