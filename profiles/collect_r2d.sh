set -x
cd /root/repo
T=r2d
(timeout 900 python -m pytest tests -m gpu -q) > gpurun_out/${T}_pytest_gpu.log 2>&1; tail -1 gpurun_out/${T}_pytest_gpu.log
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py --no-cpu-baseline --workload carbon --reps 10 10 10 --steps 40 --warmup 5 > gpurun_out/${T}_bench_carbon1m.json 2>/dev/null
python bench.py --no-cpu-baseline --workload unep --steps 30 --warmup 5 > gpurun_out/${T}_bench_unep.json 2>/dev/null
python bench.py --no-cpu-baseline --workload pbte_ortho > gpurun_out/${T}_bench_ortho.json 2>/dev/null
bash profiles/kstats.sh $T > /dev/null 2>&1
bash profiles/collect_pmc.sh $T > gpurun_out/${T}_collect_pmc.log 2>&1
python profiles/make_traffic.py gpurun_out/${T}_pmc_fetch.csv gpurun_out/${T}_pmc_write.csv 1024000 gpurun_out/${T}_traffic.json
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
python -c "import json; d=json.load(open('gpurun_out/r2d_traffic.json')); print({k:round(v['hbm_bytes_per_launch']/1e9,3) for k,v in d['kernels'].items()})"
