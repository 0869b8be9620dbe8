# size sweep of the PbTe NVE line + the Si Tersoff config-2 line: bash profiles/size_sweep.sh <tag>
cd /root/repo
T=$1
for r in "4 4 4" "8 8 8" "12 12 12" "16 16 16"; do
  n=$(echo $r | tr ' ' x)
  python bench.py --no-cpu-baseline --reps $r --steps 400 --warmup 50 > gpurun_out/sweep_${T}_$n.json 2> gpurun_out/sweep_${T}_$n.err
done
python bench.py --workload si_tersoff --steps 2000 --warmup 200 --cpu-seconds 6 > gpurun_out/sweep_${T}_si.json 2> gpurun_out/sweep_${T}_si.err
for f in gpurun_out/sweep_${T}_*.json; do
python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "atoms %d ms/step %.4f value %.4g"%(d["config"]["atoms_total"], d["ms_per_step"], d["value"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items()}, (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
