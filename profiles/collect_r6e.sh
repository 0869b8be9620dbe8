# round 6e: cover shapes (zero-padded models), by-type matrix-core ANN over flat segments
set -x
cd /root/repo
T=r6e
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_jit_shapes.py -m gpu -q -x) > gpurun_out/pytest_${T}.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_${T}.log | head
for s in 1 0; do
NEPMI_BENCH_MFMA=$s timeout 300 python bench.py --no-cpu-baseline --no-extras --workload unep --reps 16 16 16 --steps 20 --warmup 5 > gpurun_out/bench_${T}_u_m$s.json 2> gpurun_out/bench_${T}_u_m$s.err
python - gpurun_out/bench_${T}_u_m$s.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("unep ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
# PbTe 1 M atoms: own shape / padded into the cover shape / run-time shape (NEPMI_JIT=0 + generic is the bench extra)
for c in 0 1; do
NEPMI_JIT=2 NEPMI_FORCE_COVER=$c timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > gpurun_out/bench_${T}_p_c$c.json 2> gpurun_out/bench_${T}_p_c$c.err
python - gpurun_out/bench_${T}_p_c$c.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("pbte ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
    print(d["config"]["kernel_forms"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
# C_2024 512k: JIT core / cover (no core: hide it) 
NEPMI_JIT=2 timeout 300 python bench.py --no-cpu-baseline --no-extras --workload carbon2024 --reps 10 10 10 --steps 10 --warmup 2 > gpurun_out/bench_${T}_c24_jit.json 2> gpurun_out/bench_${T}_c24_jit.err
mkdir -p /tmp/hide && mv gpumd_amd/lib/jit/libnepmi_jit_12_16_8_12_6_1_* /tmp/hide/
NEPMI_JIT=2 timeout 300 python bench.py --no-cpu-baseline --no-extras --workload carbon2024 --reps 10 10 10 --steps 10 --warmup 2 > gpurun_out/bench_${T}_c24_cover.json 2> gpurun_out/bench_${T}_c24_cover.err
mv /tmp/hide/* gpumd_amd/lib/jit/
for f in gpurun_out/bench_${T}_c24_jit.json gpurun_out/bench_${T}_c24_cover.json; do
python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("c2024 ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
    print(d["config"]["kernel_forms"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
