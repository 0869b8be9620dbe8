set -x
cd /root/repo
cp gpumd_amd/lib/libnepmi.so /tmp/libnepmi_product.so
run() { # name, bench args
  name=$1; shift
  timeout 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/r3m_$name.json 2> gpurun_out/r3m_$name.err
  python - gpurun_out/r3m_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-16s ms/step %.5f value %.4g"%(sys.argv[2], d["ms_per_step"], d["value"]), {k[:12]:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k!="list_rebuild"})
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
run prod_pbte
run prod_carbon --workload carbon --reps 10 10 10 --steps 30 --warmup 5
for v in rw5 rw6; do
  cp gpumd_amd/lib/variants/libnepmi_$v.so gpumd_amd/lib/libnepmi.so
  run ${v}_pbte
  if [ $v = rw5 ]; then run ${v}_carbon --workload carbon --reps 10 10 10 --steps 30 --warmup 5; fi
done
cp /tmp/libnepmi_product.so gpumd_amd/lib/libnepmi.so
timeout 300 python profiles/inproc_weak.py --ranks 2 --steps 100 > gpurun_out/r3m_inproc_weak.json 2> gpurun_out/r3m_inproc_weak.err; tail -3 gpurun_out/r3m_inproc_weak.json | cut -c1-600
