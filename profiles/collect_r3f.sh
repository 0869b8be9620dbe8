# round 3, sixth GPU call: many-type force assembly from the neighbours' Fp rows (UNEP), carbon ANN without scratch, the bench
# line with its extra measurements
set -x
cd /root/repo
cp gpumd_amd/lib/libnepmi.so /tmp/libnepmi_product.so
run() { # name, bench args
  name=$1; shift
  timeout 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/r3f_$name.json 2> gpurun_out/r3f_$name.err
  python - gpurun_out/r3f_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-16s ms/step %.4f"%(sys.argv[2], d["ms_per_step"]), {k[:12]:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k!="list_rebuild"})
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
(timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_full_size_parity.py tests/test_ref_md_parity.py -m gpu -q -x -k "window_layouts or UNEP or unep or C-262k") > gpurun_out/r3f_pytest.log 2>&1; grep -E "passed|failed|Error" gpurun_out/r3f_pytest.log | tail -5
run prod_unep --workload unep --steps 20 --warmup 5
run prod_carbon --workload carbon --reps 10 10 10 --steps 30 --warmup 5
cp gpumd_amd/lib/variants/libnepmi_nofpj.so gpumd_amd/lib/libnepmi.so
run nofpj_unep --workload unep --steps 20 --warmup 5
cp /tmp/libnepmi_product.so gpumd_amd/lib/libnepmi.so
timeout 400 python bench.py > gpurun_out/r3f_bench.json 2> gpurun_out/r3f_bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3f_bench.json").read().strip().splitlines()[-1])
print("bench ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), d["roofline"]["frac"], d["config"]["kernel_forms"])
print(json.dumps(d.get("extra_measurements"), indent=0)[:1500])
PY
