# round 3, third GPU call: A/B of the static window layout (win2), packed compact list (cword) and occupancy caps
set -x
cd /root/repo
cp gpumd_amd/lib/libnepmi.so /tmp/libnepmi_product.so
run() { # name, bench args
  name=$1; shift
  timeout 150 python bench.py --no-cpu-baseline "$@" > gpurun_out/r3c_$name.json 2> gpurun_out/r3c_$name.err
  python - gpurun_out/r3c_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-16s ms/step %.4f"%(sys.argv[2], d["ms_per_step"]), {k[:12]:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k!="list_rebuild"})
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
run prod_pbte
run prod_carbon --workload carbon --reps 10 10 10 --steps 30 --warmup 5
run prod_unep --workload unep --steps 20 --warmup 5
for v in old w2c0 w2w5 w2w6; do
  cp gpumd_amd/lib/variants/libnepmi_$v.so gpumd_amd/lib/libnepmi.so
  run ${v}_pbte
  if [ $v = old ] || [ $v = w2w5 ]; then
    run ${v}_carbon --workload carbon --reps 10 10 10 --steps 30 --warmup 5
    run ${v}_unep --workload unep --steps 20 --warmup 5
  fi
done
cp /tmp/libnepmi_product.so gpumd_amd/lib/libnepmi.so
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_full_size_parity.py -m gpu -q -x -k "window_layouts or lanes_per_atom or full_size or nve_run or long_nve") > gpurun_out/r3c_pytest.log 2>&1; grep -E "passed|failed|Error" gpurun_out/r3c_pytest.log | tail -5
