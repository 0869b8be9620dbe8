# round 6c: matrix-core ANN for many-type models, late evaluation in the many-type radial pass
set -x
cd /root/repo
T=r6c
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_model_variants.py -m gpu -q -x) > gpurun_out/pytest_${T}_par.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_${T}_par.log | head -20
(timeout 900 python -m pytest tests/test_ref_md_parity.py -m gpu -q -x -s -k "scatter_form") > gpurun_out/pytest_${T}_traj.log 2>&1; grep -E "passed|failed|^E |^FAILED|MD parity" gpurun_out/pytest_${T}_traj.log | head -20
(timeout 600 python -m pytest tests/test_dist_inproc.py -m gpu -q -x -k "guard or reverse_exchange") > gpurun_out/pytest_${T}_dist.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_${T}_dist.log | head -20
for s in 1 0; do
NEPMI_BENCH_SYNC=$s NEPMI_BENCH_MFMA=$s timeout 300 python bench.py --no-cpu-baseline --no-extras --workload unep --reps 16 16 16 --steps 20 --warmup 5 > gpurun_out/bench_${T}_u_s$s.json 2> gpurun_out/bench_${T}_u_s$s.err
python - gpurun_out/bench_${T}_u_s$s.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("unep ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
    print(d["config"]["kernel_forms"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
