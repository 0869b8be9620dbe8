# round 5: one force kernel per brick (nep_brick.h) -- parity, then same-box A/B against fused angular + scatter kernels
cd /root/repo
T=r5h
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_force_kernel_per_brick or scatter or fused_angular or run_loop") > gpurun_out/${T}_pytest.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/${T}_pytest.log | head -20
for v in 1 0 1 0; do
NEPMI_BENCH_BRICK=$v python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10 > gpurun_out/${T}_bench_b$v.json 2> gpurun_out/${T}_bench_b$v.err
python - gpurun_out/${T}_bench_b$v.json $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("brick=%s ms/step %.4f value %.4g"%(sys.argv[2], d["ms_per_step"], d["value"]), {k[:12]:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
