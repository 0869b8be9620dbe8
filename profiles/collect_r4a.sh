# round 4, first GPU call: the scatter-form force assembly (nep_scatter.h) -- parity tests, A/B against the gather form on
# PbTe 1 M and carbon 1 M, kernel stats; bench.py's self-launch test.
set -x
cd /root/repo
T=r4a
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scatter" > gpurun_out/${T}_pytest_scatter.log 2>&1
tail -15 gpurun_out/${T}_pytest_scatter.log
for form in 0 1; do
  NEPMI_BENCH_FORM=$form timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10 > gpurun_out/${T}_bench_form$form.json 2> gpurun_out/${T}_bench_form$form.err
  NEPMI_BENCH_FORM=$form timeout 200 python bench.py --no-cpu-baseline --no-extras --workload carbon --reps 10 10 10 --steps 20 --warmup 5 > gpurun_out/${T}_bench_carbon_form$form.json 2>/dev/null
  NEPMI_BENCH_FORM=$form timeout 200 python bench.py --no-cpu-baseline --no-extras --workload pbte_ortho --steps 40 --warmup 10 > gpurun_out/${T}_bench_ortho_form$form.json 2>/dev/null
done
kst() { tag=$1; shift
  cd /tmp && export TMPDIR=/tmp
  timeout 240 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_$tag -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-extras "$@" > /root/repo/gpurun_out/prof_$tag.log 2>&1
  cd /root/repo
  python profiles/summarize_rocpd.py stats $(ls gpurun_out/prof_$tag/*.db | head -1) gpurun_out/${tag}_kernel_stats.csv
  rm -rf gpurun_out/prof_$tag
  head -12 gpurun_out/${tag}_kernel_stats.csv | cut -c1-200
}
kst ${T} --steps 20 --warmup 5
kst ${T}_carbon --workload carbon --reps 10 10 10 --steps 20 --warmup 5
timeout 600 python -m pytest tests/test_bench_launch.py -x -q -m gpu > gpurun_out/${T}_pytest_launch.log 2>&1
tail -15 gpurun_out/${T}_pytest_launch.log
for f in gpurun_out/${T}_bench*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()}, d["config"].get("kernel_forms","")[-70:])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
