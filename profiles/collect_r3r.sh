#!/bin/bash
# r3r: where the time of an 8-rank in-process strong-scaling step goes (kernel traces of both ghost modes)
set -x
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
for g in 0 1; do
  rm -rf /tmp/tr$g
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$g -- python profiles/inproc_weak.py --strong --ranks 8 --steps 40 --warmup 10 --ghosts $g --only ranks > gpurun_out/r3r_strong8_g$g.json 2> gpurun_out/r3r_strong8_g$g.err
  cat gpurun_out/r3r_strong8_g$g.json
  f=$(find /tmp/tr$g -name "*kernel_trace.csv" | head -1)
  python profiles/tools/trace_union.py $f --last-ms 150 > gpurun_out/r3r_strong8_g${g}_trace.txt 2>&1
  cat gpurun_out/r3r_strong8_g${g}_trace.txt
done
rm -rf /tmp/tr1d
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1d -- python profiles/inproc_weak.py --strong --ranks 8 --steps 40 --warmup 10 --only one > gpurun_out/r3r_strong_one.json 2>/dev/null
python profiles/tools/trace_union.py $(find /tmp/tr1d -name "*kernel_trace.csv" | head -1) --last-ms 60 > gpurun_out/r3r_strong_one_trace.txt 2>&1; cat gpurun_out/r3r_strong_one_trace.txt
timeout 600 python -m pytest tests/test_dist.py -m gpu -q -x -k reverse > gpurun_out/r3r_pytest_reverse.log 2>&1; tail -3 gpurun_out/r3r_pytest_reverse.log
