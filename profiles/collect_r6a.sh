# round 6a: the per-step radial list as wave-synchronous words (SyncFifo) -- parity subset, then same-box A/B against the slot-major compact list
set -x
cd /root/repo
T=r6a
(timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x) > gpurun_out/pytest_gpu_$T.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_gpu_$T.log | head -20
for s in 1 0 1 0; do
NEPMI_BENCH_SYNC=$s python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10 > gpurun_out/bench_${T}_s$s.json 2> gpurun_out/bench_${T}_s$s.err
python - gpurun_out/bench_${T}_s$s.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
for s in 1 0; do
NEPMI_BENCH_SYNC=$s python bench.py --no-cpu-baseline --no-extras --workload carbon --reps 10 10 10 --steps 40 --warmup 5 > gpurun_out/bench_${T}_c_s$s.json 2> gpurun_out/bench_${T}_c_s$s.err
python - gpurun_out/bench_${T}_c_s$s.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("carbon ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
