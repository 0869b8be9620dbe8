# round 4, final tree: GPU tier, smoke(), the driver's bench command
cd /root/repo
timeout 1400 python -m pytest tests -m gpu -x -q > gpurun_out/r4aq_pytest_gpu_full.log 2>&1; echo pytest rc=$?
grep -E "passed|failed|^\.+" gpurun_out/r4aq_pytest_gpu_full.log | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4aq_bench.json 2> gpurun_out/r4aq_bench.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4aq_bench.json').read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], round(d["roofline"]["frac"],3))
print({k:(round(v["avg_ms"],4), round(v.get("frac",0),3)) for k,v in d["kernels"].items()})
print({k:(v.get("ms_per_step"), v.get("value")) for k,v in d["extra_measurements"].items()})
print(d["cpu_baseline"]["value"], d["cpu_baseline"].get("reference_gpu_same_box",{}).get("value"))
PY
