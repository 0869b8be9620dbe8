set -x
cd /root/repo
T=r2a
(timeout 1200 python -m pytest tests -m gpu -q -x) > gpurun_out/pytest_gpu_$T.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_gpu_$T.log | head -20
python bench.py > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err; tail -c 600 gpurun_out/bench_$T.err
for r in "4 4 4" "8 8 8"; do python bench.py --reps $r --steps 400 --warmup 50 --no-cpu-baseline > "gpurun_out/bench_${T}_reps$(echo $r | tr -d ' ').json" 2>/dev/null; done
python bench.py --workload carbon --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_${T}_carbon.json 2> gpurun_out/bench_${T}_carbon.err
python bench.py --workload unep --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_${T}_unep.json 2> gpurun_out/bench_${T}_unep.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_$T -o bench -- python /root/repo/bench.py --no-cpu-baseline > /root/repo/gpurun_out/prof_$T.log 2>&1
cd /root/repo
python profiles/summarize_rocpd.py stats $(ls gpurun_out/prof_$T/*.db | head -1) gpurun_out/${T}_kernel_stats.csv
head -12 gpurun_out/${T}_kernel_stats.csv
for f in gpurun_out/bench_${T}*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["config"]["workload"][:40], "ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
rm -rf gpurun_out/prof_$T
