set -x
cd /root/repo
cp gpumd_amd/lib/libnepmi.so /tmp/libnepmi_product.so
run() { # name, bench args
  name=$1; shift
  timeout 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/r3h_$name.json 2> gpurun_out/r3h_$name.err
  python - gpurun_out/r3h_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-16s ms/step %.5f value %.4g"%(sys.argv[2], d["ms_per_step"], d["value"]), {k[:12]:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k!="list_rebuild"})
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
(timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_full_size_parity.py tests/test_ref_md_parity.py tests/test_tersoff.py tests/test_host_active.py tests/test_host_cli.py -m gpu -q -x -k "window_layouts or UNEP or unep or tersoff or Tersoff or active or carbon") > gpurun_out/r3h_pytest.log 2>&1; grep -E "passed|failed|Error" gpurun_out/r3h_pytest.log | tail -5
run unep --workload unep --steps 20 --warmup 5
run si --workload si_tersoff --steps 2000 --warmup 200
cp gpumd_amd/lib/variants/libnepmi_ctscalar.so gpumd_amd/lib/libnepmi.so
run ctscalar_unep --workload unep --steps 20 --warmup 5
cp /tmp/libnepmi_product.so gpumd_amd/lib/libnepmi.so
