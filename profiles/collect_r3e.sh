# round 3, fifth GPU call: software-pipelined force assembly (list words 3 ahead, rows one evaluation ahead) at 4 / 3 / 2 waves
set -x
cd /root/repo
cp gpumd_amd/lib/libnepmi.so /tmp/libnepmi_product.so
run() { # name, bench args
  name=$1; shift
  timeout 150 python bench.py --no-cpu-baseline "$@" > gpurun_out/r3e_$name.json 2> gpurun_out/r3e_$name.err
  python - gpurun_out/r3e_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-16s ms/step %.4f"%(sys.argv[2], d["ms_per_step"]), {k[:12]:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k!="list_rebuild"})
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
(timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_full_size_parity.py -m gpu -q -x -k "window_layouts or full_size_parity or nve_run") > gpurun_out/r3e_pytest.log 2>&1; grep -E "passed|failed|Error" gpurun_out/r3e_pytest.log | tail -5
run prod_pbte
run prod_carbon --workload carbon --reps 10 10 10 --steps 30 --warmup 5
for v in fw3 ah2w3 fw2 rowspad; do
  cp gpumd_amd/lib/variants/libnepmi_$v.so gpumd_amd/lib/libnepmi.so
  run ${v}_pbte
  if [ $v = fw2 ]; then run ${v}_carbon --workload carbon --reps 10 10 10 --steps 30 --warmup 5; fi
done
cp /tmp/libnepmi_product.so gpumd_amd/lib/libnepmi.so
