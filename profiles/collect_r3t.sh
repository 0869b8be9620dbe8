#!/bin/bash
# r3t: the in-process multi-rank measurements with ONE hardware queue (GPU_MAX_HW_QUEUES=1): the ranks' kernels no longer
# overlap, so their durations are solo durations and their sum is the GPU work of the decomposed step
set -x
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=1
run() { # name, args...
  name=$1; shift
  rm -rf /tmp/tq_$name
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tq_$name -- python profiles/inproc_weak.py "$@" > gpurun_out/r3t_$name.json 2> gpurun_out/r3t_$name.err
  cut -c1-200 gpurun_out/r3t_$name.json
  python profiles/tools/trace_union.py $(find /tmp/tq_$name -name "*kernel_trace.csv" | head -1) --last-ms 120 --top 12 > gpurun_out/r3t_${name}_trace.txt 2>&1
  cat gpurun_out/r3t_${name}_trace.txt
}
run weak2_ranks --ranks 2 --steps 60 --warmup 20 --only ranks
run weak2_one --ranks 2 --steps 60 --warmup 20 --only one
run strong8_g0 --strong --ranks 8 --steps 40 --warmup 10 --ghosts 0 --only ranks
run strong8_g1 --strong --ranks 8 --steps 40 --warmup 10 --ghosts 1 --only ranks
