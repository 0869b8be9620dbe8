# round 6f: whole GPU tier on the tree with cover shapes + the many-type scatter behind the fused kernel
set -x
cd /root/repo
T=r6f
(timeout 1500 python -m pytest tests -m gpu -q -x) > gpurun_out/pytest_${T}.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_${T}.log | head
NEPMI_JIT=2 NEPMI_FORCE_COVER=1 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > gpurun_out/bench_${T}_p_c1.json 2> gpurun_out/bench_${T}_p_c1.err
python - gpurun_out/bench_${T}_p_c1.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("pbte-cover ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
    print(d["config"]["kernel_forms"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
