#!/bin/bash
# r3s: where the time of a 2-rank in-process weak-scaling step goes (kernel traces: two ranks, one domain of the same atoms)
set -x
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
for leg in ranks one; do
  rm -rf /tmp/trw_$leg
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trw_$leg -- python profiles/inproc_weak.py --ranks 2 --steps 60 --warmup 20 --only $leg > gpurun_out/r3s_weak2_$leg.json 2> gpurun_out/r3s_weak2_$leg.err
  cat gpurun_out/r3s_weak2_$leg.json
  python profiles/tools/trace_union.py $(find /tmp/trw_$leg -name "*kernel_trace.csv" | head -1) --last-ms 150 --top 20 > gpurun_out/r3s_weak2_${leg}_trace.txt 2>&1
  cat gpurun_out/r3s_weak2_${leg}_trace.txt
done
timeout 900 python -m pytest tests/test_dist.py tests/test_dist_inproc.py -m gpu -q -x > gpurun_out/r3s_pytest_dist.log 2>&1; tail -3 gpurun_out/r3s_pytest_dist.log
