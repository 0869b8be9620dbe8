# Round-2 profile set of the committed tree: bash profiles/collect_r2.sh <tag>   (run through gpurun; ~6 GPU-minutes)
# -> gpurun_out/<tag>_*: full GPU test log, bench lines (PbTe default incl. CPU baseline, ortho, carbon, UNEP, Si Tersoff,
#    size sweep, decomposed single rank, 2 ranks on one GPU over TCP), rocprofv3 kernel stats, PMC passes, traffic json
set -x
cd /root/repo
T=$1
(timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/${T}_pytest_gpu.log 2>&1; tail -1 gpurun_out/${T}_pytest_gpu.log
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py --no-cpu-baseline --workload pbte_ortho > gpurun_out/${T}_bench_ortho.json 2>/dev/null
python bench.py --no-cpu-baseline --workload carbon --reps 10 10 10 --steps 40 --warmup 5 > gpurun_out/${T}_bench_carbon1m.json 2>/dev/null
python bench.py --no-cpu-baseline --workload carbon --steps 20 --warmup 3 > gpurun_out/${T}_bench_carbon4m.json 2>/dev/null
python bench.py --no-cpu-baseline --workload unep --steps 30 --warmup 5 > gpurun_out/${T}_bench_unep.json 2>/dev/null
python bench.py --no-cpu-baseline --decomposed > gpurun_out/${T}_bench_decomposed1.json 2>/dev/null
NEPMI_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 50 --warmup 5 --reps 8 8 8 --no-cpu-baseline > gpurun_out/${T}_bench_2ranks_tcp.json 2> gpurun_out/${T}_bench_2ranks_tcp.err
bash profiles/size_sweep.sh $T > gpurun_out/${T}_size_sweep.txt 2>&1
bash profiles/kstats.sh $T > /dev/null 2>&1
bash profiles/kstats.sh ${T}_carbon --workload carbon --reps 10 10 10 --steps 20 --warmup 3 > /dev/null 2>&1
bash profiles/collect_pmc.sh $T > gpurun_out/${T}_collect_pmc.log 2>&1
python profiles/make_traffic.py gpurun_out/${T}_pmc_fetch.csv gpurun_out/${T}_pmc_write.csv 1024000 gpurun_out/${T}_traffic.json
for f in gpurun_out/${T}_bench*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["config"]["workload"][:36], "ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k[:8]:round(v["avg_ms"],3) for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
cat gpurun_out/${T}_size_sweep.txt | grep atoms
head -12 gpurun_out/${T}_kernel_stats.csv | cut -c1-160
