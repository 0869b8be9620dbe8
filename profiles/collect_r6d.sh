cd /root/repo
T=r6d
(timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mfma or UNEP") > gpurun_out/pytest_${T}.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_${T}.log | head
for s in 1 0; do
NEPMI_BENCH_MFMA=$s timeout 300 python bench.py --no-cpu-baseline --no-extras --workload unep --reps 16 16 16 --steps 20 --warmup 5 > gpurun_out/bench_${T}_u_m$s.json 2> gpurun_out/bench_${T}_u_m$s.err
python - gpurun_out/bench_${T}_u_m$s.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("unep ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
