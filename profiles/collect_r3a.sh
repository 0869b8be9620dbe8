# round 3, first GPU call: the whole GPU test tier (new: full-size parity, executed drop-in, oracle check of the decomposed run),
# baseline bench lines of the three model families and their kernel stats
set -x
cd /root/repo
T=r3a
(timeout 1500 python -m pytest tests -m gpu -q -x --durations=15) > gpurun_out/${T}_pytest_gpu.log 2>&1; tail -25 gpurun_out/${T}_pytest_gpu.log
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py --no-cpu-baseline --workload carbon --reps 10 10 10 --steps 40 --warmup 5 > gpurun_out/${T}_bench_carbon1m.json 2>/dev/null
python bench.py --no-cpu-baseline --workload unep --steps 30 --warmup 5 > gpurun_out/${T}_bench_unep.json 2>/dev/null
bash profiles/kstats.sh $T > /dev/null 2>&1
bash profiles/kstats.sh ${T}_carbon --workload carbon --reps 10 10 10 --steps 20 --warmup 5 > /dev/null 2>&1
bash profiles/kstats.sh ${T}_unep --workload unep --steps 20 --warmup 5 > /dev/null 2>&1
for f in gpurun_out/${T}_bench*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
head -8 gpurun_out/${T}_kernel_stats.csv | cut -c1-150
