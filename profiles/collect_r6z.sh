# round 6, final profile set: the driver's bench command, kernel stats of the three families, PMC passes (PbTe: SQ / LDS / TA / HBM-side
# traffic; UNEP and carbon: HBM-side traffic; UNEP: matrix-core counters), small sizes, the 8 M-atom carbon NVT run of config 5's per-GPU
# share, the in-process multi-rank measurements.  Every command under its own timeout.
set -x
cd /root/repo
T=r6z
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
timeout 200 python bench.py --no-cpu-baseline --no-extras --workload unep --steps 20 --warmup 5 > gpurun_out/${T}_bench_unep.json 2>/dev/null
timeout 200 python bench.py --no-cpu-baseline --no-extras --workload pbte_ortho > gpurun_out/${T}_bench_ortho.json 2>/dev/null
timeout 100 python bench.py --no-cpu-baseline --no-extras --reps 4 4 4 --steps 400 --warmup 40 > gpurun_out/${T}_bench_16k.json 2>/dev/null
timeout 100 python bench.py --no-cpu-baseline --no-extras --reps 8 8 8 --steps 200 --warmup 20 > gpurun_out/${T}_bench_128k.json 2>/dev/null
timeout 100 python bench.py --no-cpu-baseline --no-extras --reps 10 10 10 --steps 200 --warmup 20 > gpurun_out/${T}_bench_250k.json 2>/dev/null
timeout 200 python bench.py --no-cpu-baseline --no-extras --workload carbon --reps 10 10 10 --steps 30 --warmup 5 > gpurun_out/${T}_bench_carbon1m.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --workload carbon --reps 20 20 20 --decomposed --ensemble nvt_ber --steps 20 --warmup 5 > gpurun_out/${T}_bench_carbon8m_nvt.json 2> gpurun_out/${T}_bench_carbon8m_nvt.err
timeout 200 python bench.py --no-cpu-baseline --no-extras --decomposed --steps 40 --warmup 10 > gpurun_out/${T}_bench_decomposed1.json 2>/dev/null
kst() { tag=$1; shift
  cd /tmp && export TMPDIR=/tmp
  timeout 240 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_$tag -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-extras "$@" > /root/repo/gpurun_out/prof_$tag.log 2>&1
  cd /root/repo
  python profiles/summarize_rocpd.py stats $(ls gpurun_out/prof_$tag/*.db | head -1) gpurun_out/${tag}_kernel_stats.csv
  rm -rf gpurun_out/prof_$tag
  head -9 gpurun_out/${tag}_kernel_stats.csv | cut -c1-170
}
kst ${T} --steps 20 --warmup 5
kst ${T}_carbon --workload carbon --reps 10 10 10 --steps 20 --warmup 5
kst ${T}_unep --workload unep --steps 20 --warmup 5
pmc() { # tag, counters..., then -- bench args
  tag=$1; shift; ctrs=""; while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
  cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --pmc $ctrs -d /root/repo/gpurun_out/prof_$tag -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 "$@" > /root/repo/gpurun_out/prof_$tag.log 2>&1
  cd /root/repo
  python profiles/summarize_rocpd.py pmc $(ls gpurun_out/prof_$tag/*.db | head -1) gpurun_out/${tag}.csv
  rm -rf gpurun_out/prof_$tag
  grep -E "scatter_kernel|scatter_mt|ForceFold|RadialWin2|AngularFused|ResidentStep|ann_mfma" gpurun_out/${tag}.csv | cut -c1-200 | head -40
}
pmc ${T}_pmc_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --
pmc ${T}_pmc_sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT --
pmc ${T}_pmc_lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_LDS --
pmc ${T}_pmc_fetch FETCH_SIZE --
pmc ${T}_pmc_write WRITE_SIZE --
pmc ${T}_pmc_ta TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE --
pmc ${T}_unep_pmc_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -- --workload unep
pmc ${T}_unep_pmc_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA -- --workload unep
pmc ${T}_unep_pmc_fetch FETCH_SIZE -- --workload unep
pmc ${T}_unep_pmc_write WRITE_SIZE -- --workload unep
pmc ${T}_carbon_pmc_fetch FETCH_SIZE -- --workload carbon --reps 10 10 10
pmc ${T}_carbon_pmc_write WRITE_SIZE -- --workload carbon --reps 10 10 10
rm -f gpurun_out/${T}_traffic_all.json
python profiles/make_traffic.py gpurun_out/${T}_pmc_fetch.csv gpurun_out/${T}_pmc_write.csv 1024000 gpurun_out/${T}_traffic.json gpurun_out/${T}_traffic_all.json
python profiles/make_traffic.py gpurun_out/${T}_unep_pmc_fetch.csv gpurun_out/${T}_unep_pmc_write.csv 1048576 gpurun_out/${T}_unep_traffic.json gpurun_out/${T}_traffic_all.json
python profiles/make_traffic.py gpurun_out/${T}_carbon_pmc_fetch.csv gpurun_out/${T}_carbon_pmc_write.csv 1000000 gpurun_out/${T}_carbon_traffic.json gpurun_out/${T}_traffic_all.json
for f in gpurun_out/${T}_bench*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()}, d.get("device_memory"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
python -c "import json; d=json.load(open('gpurun_out/r6z_traffic_all.json')); print({a:{k:round(v['hbm_bytes_per_launch']/1e9,3) for k,v in e['kernels'].items()} for a,e in d['by_atoms'].items()})"
timeout 300 python profiles/inproc_weak.py --ranks 2 --steps 100 > gpurun_out/${T}_inproc_weak2.json 2>/dev/null; cut -c1-330 gpurun_out/${T}_inproc_weak2.json
timeout 300 python profiles/inproc_weak.py --ranks 4 --steps 60 > gpurun_out/${T}_inproc_weak4.json 2>/dev/null; cut -c1-330 gpurun_out/${T}_inproc_weak4.json
for g in 0 1; do
  timeout 300 python profiles/inproc_weak.py --strong --ranks 8 --steps 60 --ghosts $g > gpurun_out/${T}_inproc_strong8_g${g}.json 2>/dev/null; grep -o '"ms_per_step_ranks[^,]*, "ms_per_step_one[^,]*, "work_inflation[^,]*' gpurun_out/${T}_inproc_strong8_g${g}.json
done
