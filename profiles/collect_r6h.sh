cd /root/repo
(timeout 600 python -m pytest tests/test_tersoff.py tests/test_ref_md_parity.py -m gpu -q -x -k "tersoff or Tersoff or si_") > gpurun_out/pytest_r6h.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_r6h.log | head
timeout 200 python bench.py --no-cpu-baseline --no-extras --workload si_tersoff --steps 2000 --warmup 200 > gpurun_out/bench_r6h_si.json 2> gpurun_out/bench_r6h_si.err
python - gpurun_out/bench_r6h_si.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("si ms/step %.5f value %.4g"%(d["ms_per_step"], d["value"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
