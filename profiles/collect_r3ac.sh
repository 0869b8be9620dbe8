#!/bin/bash
# r3ac: rebuild kernels after the binary-search reverse slots (kernel stats of one rebuild per family) + GPU parity
set -x
cd /root/repo; mkdir -p gpurun_out
kst() { tag=$1; shift
  cd /tmp && export TMPDIR=/tmp
  timeout 240 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_$tag -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-extras "$@" > /root/repo/gpurun_out/prof_$tag.log 2>&1
  cd /root/repo
  python profiles/summarize_rocpd.py stats $(ls gpurun_out/prof_$tag/*.db | head -1) gpurun_out/${tag}_kernel_stats.csv
  rm -rf gpurun_out/prof_$tag
  grep -E "ReverseSlots|BuildLists|PackCodes|TileStats" gpurun_out/${tag}_kernel_stats.csv | cut -c1-110
}
kst r3ac --steps 10 --warmup 2
kst r3ac_carbon --workload carbon --reps 10 10 10 --steps 10 --warmup 2
kst r3ac_unep --workload unep --steps 10 --warmup 2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size_parity.py tests/test_dist.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED" | tail -3
