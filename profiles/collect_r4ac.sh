# round 4: what binds the many-type radial pass and scatter (UNEP-v1): LDS counters
set -x
cd /root/repo
T=r4ac
pmc() { # tag, counters..., then -- bench args
  tag=$1; shift; ctrs=""; while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
  cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --pmc $ctrs -d /root/repo/gpurun_out/prof_$tag -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 "$@" > /root/repo/gpurun_out/prof_$tag.log 2>&1
  cd /root/repo
  python profiles/summarize_rocpd.py pmc $(ls gpurun_out/prof_$tag/*.db | head -1) gpurun_out/${tag}.csv
  rm -rf gpurun_out/prof_$tag
  grep -E "scatter|ForceFold|RadialWin2|AngularForce|AngularDesc|AnnBody" gpurun_out/${tag}.csv | cut -c1-40,150-260 | head -60
}
pmc ${T}_unep_pmc_lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -- --workload unep
pmc ${T}_unep_pmc_sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES -- --workload unep
