# round 6j: angular records of list A as wave-synchronous rows -- parity (whole tier), then same-box A/B (NEPMI_BENCH_SYNC=1 / 0 switches both list forms)
cd /root/repo
(timeout 1500 python -m pytest tests -m gpu -q -x) > gpurun_out/pytest_r6j.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_r6j.log | head
for w in pbte carbon unep; do
for s in 1 0; do
if [ $w = pbte ]; then A="--steps 40 --warmup 10"; elif [ $w = carbon ]; then A="--workload carbon --reps 10 10 10 --steps 30 --warmup 5"; else A="--workload unep --reps 16 16 16 --steps 20 --warmup 5"; fi
NEPMI_BENCH_SYNC=$s timeout 300 python bench.py --no-cpu-baseline --no-extras $A > gpurun_out/bench_r6j_${w}_s$s.json 2> gpurun_out/bench_r6j_${w}_s$s.err
python - gpurun_out/bench_r6j_${w}_s$s.json $w$s <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items()}, "nn_a", round(d["config"]["mean_nn_angular"],3))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
done
