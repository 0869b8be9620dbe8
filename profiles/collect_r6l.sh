# round 6l: the size rule again with the round-6 kernels -- one lane per atom + scatter form forced at small sizes against the rule's choice
cd /root/repo
for r in "4 4 4" "6 6 6" "8 8 8" "9 9 9"; do
for v in rule one_lane_scatter one_lane_gather; do
E=""; [ $v = one_lane_scatter ] && E="NEPMI_BENCH_LANES=1 NEPMI_BENCH_FORM=1"; [ $v = one_lane_gather ] && E="NEPMI_BENCH_LANES=1 NEPMI_BENCH_FORM=0"
env $E timeout 120 python bench.py --no-cpu-baseline --no-extras --reps $r --steps 300 --warmup 40 > gpurun_out/bench_r6l.json 2>/dev/null
python - gpurun_out/bench_r6l.json "$r $v" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k[:10]:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k!="list_rebuild"}, d["config"]["kernel_forms"].split("window=")[1][:40])
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
done
done
