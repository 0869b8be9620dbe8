#!/bin/bash
# r3q: reverse-mode ghosts (nepmi_dist_set_ghost_mode): GPU tests of the decomposed driver, strong and weak in-process measurements
set -x
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dist.py tests/test_dist_inproc.py -m gpu -q -x > gpurun_out/r3q_pytest_dist.log 2>&1; tail -3 gpurun_out/r3q_pytest_dist.log
for g in 0 1; do
  timeout 300 python profiles/inproc_weak.py --strong --ranks 8 --steps 60 --ghosts $g > gpurun_out/r3q_inproc_strong8_g$g.json 2> gpurun_out/r3q_inproc_strong8_g$g.err; cat gpurun_out/r3q_inproc_strong8_g$g.json
  timeout 300 python profiles/inproc_weak.py --ranks 2 --steps 100 --ghosts $g > gpurun_out/r3q_inproc_weak2_g$g.json 2> gpurun_out/r3q_inproc_weak2_g$g.err; cat gpurun_out/r3q_inproc_weak2_g$g.json
done
timeout 300 python profiles/inproc_weak.py --strong --ranks 2 --steps 60 --ghosts 1 > gpurun_out/r3q_inproc_strong2_g1.json 2>/dev/null; cat gpurun_out/r3q_inproc_strong2_g1.json
timeout 300 python profiles/inproc_weak.py --strong --ranks 4 --steps 60 --ghosts 1 > gpurun_out/r3q_inproc_strong4_g1.json 2>/dev/null; cat gpurun_out/r3q_inproc_strong4_g1.json
