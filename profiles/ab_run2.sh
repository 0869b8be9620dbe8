# like ab_run.sh, for a second variants directory: bash profiles/ab_run2.sh <dir> <tag>
cd /root/repo
cp gpumd_amd/lib/libnepmi.so /tmp/libnepmi_product.so
for v in gpumd_amd/lib/$1/libnepmi_*.so; do
  name=$(basename $v .so); name=${name#libnepmi_}
  cp $v gpumd_amd/lib/libnepmi.so
  python bench.py --no-cpu-baseline $AB_ARGS > gpurun_out/ab_$2_$name.json 2> gpurun_out/ab_$2_$name.err
  python - gpurun_out/ab_$2_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-12s ms/step %.4f"%(sys.argv[2], d["ms_per_step"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k!="list_rebuild"})
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
done
cp /tmp/libnepmi_product.so gpumd_amd/lib/libnepmi.so
