cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
kst() { tag=$1; shift
  cd /tmp && export TMPDIR=/tmp
  timeout 240 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_$tag -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-extras "$@" > /root/repo/gpurun_out/prof_$tag.log 2>&1
  cd /root/repo
  python profiles/summarize_rocpd.py stats $(ls gpurun_out/prof_$tag/*.db | head -1) gpurun_out/${tag}_kernel_stats.csv
  rm -rf gpurun_out/prof_$tag
  grep -E "BuildLists|PackCodes|ReverseSlots|TileStats|FoldMap" gpurun_out/${tag}_kernel_stats.csv | cut -c1-120
}
kst r4ap_rebuild --steps 100 --warmup 10
kst r4ap_rebuild_carbon --workload carbon --reps 10 10 10 --steps 30 --warmup 5
kst r4ap_rebuild_unep --workload unep --steps 60 --warmup 10
