# GPU box: the decomposed driver on one GPU -- tests (ranks share the GPU over TCP), overhead of the dist path on one rank,
# and 2-rank sanity lines for the three model families
set -x
cd /root/repo
T=$1
(timeout 1200 python -m pytest tests/test_dist.py -m gpu -q -x) > gpurun_out/pytest_dist_$T.log 2>&1; grep -E "passed|failed|^E |^FAILED|Error" gpurun_out/pytest_dist_$T.log | head -20
python bench.py --no-cpu-baseline > gpurun_out/bench_${T}_plain.json 2> gpurun_out/bench_${T}_plain.err
python bench.py --no-cpu-baseline --decomposed > gpurun_out/bench_${T}_dec1.json 2> gpurun_out/bench_${T}_dec1.err
for w in pbte carbon unep; do
  NEPMI_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload $w --reps 6 6 6 --steps 20 --warmup 5 --ensemble nvt_ber > gpurun_out/bench_${T}_2r_$w.json 2> gpurun_out/bench_${T}_2r_$w.err
done
for f in gpurun_out/bench_${T}_plain.json gpurun_out/bench_${T}_dec1.json gpurun_out/bench_${T}_2r_pbte.json gpurun_out/bench_${T}_2r_carbon.json gpurun_out/bench_${T}_2r_unep.json; do
python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "ms/step %.4f value %.4g T=%.1f"%(d["ms_per_step"], d["value"], d["thermo_last"][0]), d["config"].get("parallelism","")[:60], d["config"].get("decompositions_in_timed_region"))
except Exception as e:
    print(sys.argv[1], "ERR", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
