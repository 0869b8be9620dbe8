# usage: bash profiles/collect_pmc.sh <tag> [bench args...]   -- separate rocprofv3 --pmc passes (no tracing domains)
set -x
T=$1; shift
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --steps 6 --warmup 2 $*"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -d /root/repo/gpurun_out/prof_${T}_sq1 -o bench -- $B > /root/repo/gpurun_out/prof_${T}_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT -d /root/repo/gpurun_out/prof_${T}_sq2 -o bench -- $B > /root/repo/gpurun_out/prof_${T}_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d /root/repo/gpurun_out/prof_${T}_fetch -o bench -- $B > /root/repo/gpurun_out/prof_${T}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /root/repo/gpurun_out/prof_${T}_write -o bench -- $B > /root/repo/gpurun_out/prof_${T}_write.log 2>&1
cd /root/repo
for k in sq1 sq2 fetch write; do
  python profiles/summarize_rocpd.py pmc $(ls gpurun_out/prof_${T}_$k/*.db | head -1) gpurun_out/${T}_pmc_$k.csv
  rm -rf gpurun_out/prof_${T}_$k
done
grep -E "WinBody|AngularForce|AngularDesc|ann_mfma|ResidentStep" gpurun_out/${T}_pmc_sq1.csv | head -60
