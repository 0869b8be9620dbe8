# round 6n: the two-type cover shape (8,12,8,12,6;2) -- parity, then PbTe 1 M atoms forced onto it
cd /root/repo
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_jit_shapes.py -m gpu -q -x) > gpurun_out/pytest_r6n.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_r6n.log | head
NEPMI_JIT=2 NEPMI_FORCE_COVER=1 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > gpurun_out/bench_r6n_p_c1.json 2> gpurun_out/bench_r6n_p_c1.err
python - gpurun_out/bench_r6n_p_c1.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("pbte-cover ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
    print(d["config"]["kernel_forms"])
except Exception as e:
    print(sys.argv[1], "ERR", e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
