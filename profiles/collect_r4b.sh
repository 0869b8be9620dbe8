# round 4, second GPU call: scatter form v2 (12-byte LDS records, three workgroups per CU, halo rows + fold map): parity, A/B,
# kernel stats, two PMC passes; then the whole GPU tier (the run loops now take the scatter form by default).
set -x
cd /root/repo
T=r4b
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scatter" > gpurun_out/${T}_pytest_scatter.log 2>&1
tail -5 gpurun_out/${T}_pytest_scatter.log
for form in 0 1; do
  NEPMI_BENCH_FORM=$form timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10 > gpurun_out/${T}_bench_form$form.json 2> gpurun_out/${T}_bench_form$form.err
  NEPMI_BENCH_FORM=$form timeout 200 python bench.py --no-cpu-baseline --no-extras --workload carbon --reps 10 10 10 --steps 20 --warmup 5 > gpurun_out/${T}_bench_carbon_form$form.json 2>/dev/null
done
kst() { tag=$1; shift
  cd /tmp && export TMPDIR=/tmp
  timeout 240 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_$tag -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-extras "$@" > /root/repo/gpurun_out/prof_$tag.log 2>&1
  cd /root/repo
  python profiles/summarize_rocpd.py stats $(ls gpurun_out/prof_$tag/*.db | head -1) gpurun_out/${tag}_kernel_stats.csv
  rm -rf gpurun_out/prof_$tag
  head -8 gpurun_out/${tag}_kernel_stats.csv | cut -c1-200
}
kst ${T} --steps 20 --warmup 5
kst ${T}_carbon --workload carbon --reps 10 10 10 --steps 20 --warmup 5
pmc() { # tag, counters..., then -- bench args
  tag=$1; shift; ctrs=""; while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
  cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --pmc $ctrs -d /root/repo/gpurun_out/prof_$tag -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 "$@" > /root/repo/gpurun_out/prof_$tag.log 2>&1
  cd /root/repo
  python profiles/summarize_rocpd.py pmc $(ls gpurun_out/prof_$tag/*.db | head -1) gpurun_out/${tag}.csv
  rm -rf gpurun_out/prof_$tag
  grep -E "kernel|scatter|Fold|RadialWin" gpurun_out/${tag}.csv | cut -c1-260 | head -8
}
pmc ${T}_pmc_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --
pmc ${T}_pmc_sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT --
pmc ${T}_pmc_lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_WAIT_INST_LDS SQ_INSTS_LDS --
for f in gpurun_out/${T}_bench*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()}, d["config"].get("kernel_forms","")[-70:])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${T}_pytest_gpu.log 2>&1
tail -30 gpurun_out/${T}_pytest_gpu.log
