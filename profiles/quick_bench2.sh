set -x
cd /root/repo
T=$1
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "force_parity or nve_run or unwrapped or full_size") > gpurun_out/pytest_gpu_$T.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_gpu_$T.log | head -20
python bench.py --no-cpu-baseline > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err
python bench.py --no-cpu-baseline --workload carbon --reps 10 10 10 --steps 40 --warmup 5 > gpurun_out/bench_${T}_c.json 2> gpurun_out/bench_${T}_c.err
NEPMI_BENCH_TILES=1 python bench.py --no-cpu-baseline --workload carbon --reps 10 10 10 --steps 40 --warmup 5 > gpurun_out/bench_${T}_c_t1.json 2> gpurun_out/bench_${T}_c_t1.err
for f in gpurun_out/bench_$T.json gpurun_out/bench_${T}_c.json gpurun_out/bench_${T}_c_t1.json; do
python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
PY
done
