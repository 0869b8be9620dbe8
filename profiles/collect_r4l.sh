#!/bin/bash
# r4l: the reverse exchange overlapped with the interior bricks' force assembly (nepmi_dist_set_overlap + reverse ghosts): tests, then
# in-process strong scaling on 8 ranks with the overlap off / on, both ghost forms
set -x
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4l
timeout 1500 python -m pytest tests/test_dist.py tests/test_dist_inproc.py tests/test_bench_launch.py -x -q -m gpu > gpurun_out/${T}_pytest_dist.log 2>&1
grep -E "passed|failed" gpurun_out/${T}_pytest_dist.log; grep -E "^E  |Error" gpurun_out/${T}_pytest_dist.log | head
for ov in 0 1; do
  for g in 0 1; do
    timeout 300 python profiles/inproc_weak.py --strong --ranks 8 --steps 60 --ghosts $g --overlap $ov > gpurun_out/${T}_strong8_g${g}_ov$ov.json 2>/dev/null; echo "ghosts $g overlap $ov"; grep -o '"ms_per_step_ranks[^,]*, "ms_per_step_one[^,]*, "work_inflation[^,]*' gpurun_out/${T}_strong8_g${g}_ov$ov.json
  done
  timeout 300 python profiles/inproc_weak.py --ranks 2 --steps 100 --overlap $ov > gpurun_out/${T}_weak2_ov$ov.json 2>/dev/null; echo "weak2 overlap $ov"; grep -o '"ms_per_step_ranks[^,]*, "ms_per_step_one[^,]*, "decomposition_overhead[^,]*' gpurun_out/${T}_weak2_ov$ov.json
done
