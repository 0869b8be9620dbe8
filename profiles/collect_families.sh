# kernel stats of the other two model families (single GPU): carbon C_2022_NEP4 4,096,000 atoms, UNEP-v1 1,024,000 atoms
cd /tmp && export TMPDIR=/tmp
for w in carbon unep; do
  rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_fam_$w -o bench -- python /root/repo/bench.py --workload $w --steps 30 --warmup 6 --no-cpu-baseline > /root/repo/gpurun_out/fam_$w.json 2> /root/repo/gpurun_out/fam_$w.err
  python /root/repo/profiles/summarize_rocpd.py stats $(ls /root/repo/gpurun_out/prof_fam_$w/*.db | head -1) /root/repo/gpurun_out/r1q_${w}_kernel_stats.csv
  head -7 /root/repo/gpurun_out/r1q_${w}_kernel_stats.csv | cut -c1-150
  tail -1 /root/repo/gpurun_out/fam_$w.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'], d['value'])"
done
