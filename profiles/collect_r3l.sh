set -x
cd /root/repo
cp gpumd_amd/lib/libnepmi.so /tmp/libnepmi_product.so
run() { # name, bench args
  name=$1; shift
  timeout 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/r3l_$name.json 2> gpurun_out/r3l_$name.err
  python - gpurun_out/r3l_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-16s ms/step %.5f value %.4g"%(sys.argv[2], d["ms_per_step"], d["value"]), {k[:12]:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k!="list_rebuild"})
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
run prod_pbte
for v in fill256 fill264 fill280; do
  cp gpumd_amd/lib/variants/libnepmi_$v.so gpumd_amd/lib/libnepmi.so
  run ${v}_pbte
  if [ $v = fill264 ]; then run ${v}_carbon --workload carbon --reps 10 10 10 --steps 30 --warmup 5; run ${v}_unep --workload unep --steps 20 --warmup 5; fi
done
cp /tmp/libnepmi_product.so gpumd_amd/lib/libnepmi.so
