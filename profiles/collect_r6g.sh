# round 6: kernel statistics and HBM-side traffic of C_2024_NEP4 (512,000 atoms, JIT core) on the LDS-window kernels
cd /root/repo
export NEPMI_JIT=2
T=r6z_c2024
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_$T -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-extras --workload carbon2024 --reps 10 10 10 --steps 10 --warmup 3 > /root/repo/gpurun_out/prof_$T.log 2>&1
cd /root/repo
python profiles/summarize_rocpd.py stats $(ls gpurun_out/prof_$T/*.db | head -1) gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/prof_$T
head -8 gpurun_out/${T}_kernel_stats.csv | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  tag=${T}_pmc_$(echo $c | tr A-Z a-z | sed 's/_size//')
  cd /tmp
  timeout 200 rocprofv3 --pmc $c -d /root/repo/gpurun_out/prof_$tag -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-extras --workload carbon2024 --reps 10 10 10 --steps 6 --warmup 2 > /root/repo/gpurun_out/prof_$tag.log 2>&1
  cd /root/repo
  python profiles/summarize_rocpd.py pmc $(ls gpurun_out/prof_$tag/*.db | head -1) gpurun_out/${tag}.csv
  rm -rf gpurun_out/prof_$tag
done
cp profiles/traffic_latest.json gpurun_out/r6z_traffic_all.json
python profiles/make_traffic.py gpurun_out/${T}_pmc_fetch.csv gpurun_out/${T}_pmc_write.csv 512000 gpurun_out/${T}_traffic.json gpurun_out/r6z_traffic_all.json
python -c "import json; d=json.load(open('gpurun_out/r6z_traffic_all.json')); print({a:{k:round(v['hbm_bytes_per_launch']/1e9,3) for k,v in e['kernels'].items()} for a,e in d['by_atoms'].items()})"
