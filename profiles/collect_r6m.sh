cd /root/repo
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_md_parity.py tests/test_host_cli.py -m gpu -q -x) > gpurun_out/pytest_r6m.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_r6m.log | head
for r in "9 9 9" "10 10 10" "12 12 12"; do
timeout 120 python bench.py --no-cpu-baseline --no-extras --reps $r --steps 300 --warmup 40 > gpurun_out/bench_r6m.json 2>/dev/null
python - gpurun_out/bench_r6m.json "$r" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k!="list_rebuild"})
PY
done
