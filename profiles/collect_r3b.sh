# round 3, second GPU call: what bounds the gathers of the force assembly?  (1) microbenchmark: per-lane random 16-byte reads,
# global vs LDS; (2) TA / TCP / TCC counters of the window kernels (separate --pmc passes, no tracing domains)
set -x
cd /root/repo
./profiles/microbench/gather_rate_mb > gpurun_out/r3b_gather_rate.txt 2>&1; cat gpurun_out/r3b_gather_rate.txt
cd /tmp && export TMPDIR=/tmp
pmc() { # tag, counters..., then -- bench args
  tag=$1; shift; ctrs=""; while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
  rocprofv3 --pmc $ctrs -d /root/repo/gpurun_out/prof_$tag -o bench -- python /root/repo/bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" > /root/repo/gpurun_out/prof_$tag.log 2>&1
  python /root/repo/profiles/summarize_rocpd.py pmc $(ls /root/repo/gpurun_out/prof_$tag/*.db | head -1) /root/repo/gpurun_out/${tag}.csv
  rm -rf /root/repo/gpurun_out/prof_$tag
  grep -E "WinBody|AngularForce|AngularDesc|AnnBody" /root/repo/gpurun_out/${tag}.csv | cut -c1-260 | head -12
}
pmc r3b_pmc_ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE --
pmc r3b_pmc_ta2 TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUSY_avr --
pmc r3b_pmc_tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --
pmc r3b_pmc_tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum --
pmc r3b_pmc_tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --
pmc r3b_unep_pmc_ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE -- --workload unep
pmc r3b_unep_pmc_tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum -- --workload unep
