#!/bin/bash
# r3v: boundary bricks' radial pass on the communication stream (overlap on) against overlap off; device-transport GPU tests
set -x
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dist_inproc.py tests/test_dist.py -m gpu -q -x > gpurun_out/r3v_pytest_dist.log 2>&1; tail -3 gpurun_out/r3v_pytest_dist.log
for ov in 0 1; do
  timeout 300 python profiles/inproc_weak.py --ranks 2 --steps 100 --overlap $ov > gpurun_out/r3v_weak2_ov$ov.json 2>/dev/null; cut -c1-330 gpurun_out/r3v_weak2_ov$ov.json
  timeout 300 python profiles/inproc_weak.py --strong --ranks 8 --steps 60 --ghosts 1 --overlap $ov > gpurun_out/r3v_strong8_g1_ov$ov.json 2>/dev/null; grep -o '"ms_per_step_ranks[^,]*, "ms_per_step_one[^,]*, "work_inflation[^,]*' gpurun_out/r3v_strong8_g1_ov$ov.json
done
timeout 300 python profiles/inproc_weak.py --ranks 4 --steps 60 --overlap 1 > gpurun_out/r3v_weak4_ov1.json 2>/dev/null; cut -c1-330 gpurun_out/r3v_weak4_ov1.json
