# quick GPU check of the working tree: parity subset + PbTe bench line (+ optional carbon / unep lines)
set -x
cd /root/repo
T=$1
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_model_variants.py -m gpu -q -x) > gpurun_out/pytest_gpu_$T.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_gpu_$T.log | head -20
python bench.py --no-cpu-baseline > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err
FILES="gpurun_out/bench_$T.json"
if [ "$2" = "all" ]; then
python bench.py --no-cpu-baseline --workload carbon --reps 10 10 10 --steps 40 --warmup 5 > gpurun_out/bench_${T}_c.json 2> gpurun_out/bench_${T}_c.err
python bench.py --no-cpu-baseline --workload unep --reps 16 16 16 --steps 20 --warmup 5 > gpurun_out/bench_${T}_u.json 2> gpurun_out/bench_${T}_u.err
FILES="$FILES gpurun_out/bench_${T}_c.json gpurun_out/bench_${T}_u.json"
fi
for f in $FILES; do
python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
