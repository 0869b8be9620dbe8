set -x
cd /root/repo
(timeout 300 python -m pytest tests/test_tersoff.py tests/test_ref_md_parity.py tests/test_host_cli.py -m gpu -q -x -k "tersoff or Tersoff or si_") > gpurun_out/r3j_pytest.log 2>&1; grep -E "passed|failed|Error" gpurun_out/r3j_pytest.log | tail -5
for w in 1 2; do timeout 100 python bench.py --no-cpu-baseline --workload si_tersoff --steps 2000 --warmup 200 > gpurun_out/r3j_si$w.json 2>/dev/null; python - gpurun_out/r3j_si$w.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("si ms/step %.5f value %.4g"%(d["ms_per_step"], d["value"]), {k[:12]:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k!="list_rebuild"})
PY
done
timeout 100 python bench.py --no-cpu-baseline --workload si_tersoff --reps 64 64 64 --steps 100 --warmup 10 > gpurun_out/r3j_si884k.json 2>/dev/null; python - gpurun_out/r3j_si884k.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("si884k ms/step %.5f value %.4g"%(d["ms_per_step"], d["value"]), {k[:12]:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k!="list_rebuild"})
PY
