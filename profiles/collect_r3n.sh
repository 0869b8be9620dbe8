set -x
cd /root/repo
cp gpumd_amd/lib/libnepmi.so /tmp/libnepmi_product.so
run() { # name, bench args
  name=$1; shift
  timeout 200 python bench.py --no-cpu-baseline --no-extras "$@" > gpurun_out/r3n_$name.json 2> gpurun_out/r3n_$name.err
  python - gpurun_out/r3n_$name.json $name <<'PY'
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
cp gpumd_amd/lib/variants/libnepmi_angfirst.so gpumd_amd/lib/libnepmi.so
run angfirst_pbte
run angfirst_carbon --workload carbon --reps 10 10 10 --steps 30 --warmup 5
cp /tmp/libnepmi_product.so gpumd_amd/lib/libnepmi.so
for reps in 8 10 12; do for L in 1 2 4; do NEPMI_BENCH_LANES=$L run r${reps}_L$L --reps $reps $reps $reps --steps 200 --warmup 20; done; done
run r4_auto --reps 4 4 4 --steps 400 --warmup 40
