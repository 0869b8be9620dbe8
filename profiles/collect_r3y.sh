#!/bin/bash
# r3y: forward against reverse ghosts where the per-owned-atom kernels dominate: carbon (C_2022) 1,000,000 atoms and UNEP-v1
# 864,000 atoms on 2 x 2 x 2 ranks, in process
set -x
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
for g in 0 1; do
  timeout 400 python profiles/inproc_weak.py --workload carbon --reps 50 50 50 --strong --ranks 8 --steps 30 --warmup 6 --ghosts $g > gpurun_out/r3y_carbon_strong8_g$g.json 2> gpurun_out/r3y_carbon_strong8_g$g.err; cat gpurun_out/r3y_carbon_strong8_g$g.json; tail -2 gpurun_out/r3y_carbon_strong8_g$g.err
  timeout 400 python profiles/inproc_weak.py --workload unep --reps 60 60 60 --strong --ranks 8 --steps 30 --warmup 6 --ghosts $g > gpurun_out/r3y_unep_strong8_g$g.json 2> gpurun_out/r3y_unep_strong8_g$g.err; cat gpurun_out/r3y_unep_strong8_g$g.json; tail -2 gpurun_out/r3y_unep_strong8_g$g.err
done
timeout 300 python profiles/inproc_weak.py --ranks 2 --steps 100 --ghosts 1 > gpurun_out/r3y_weak2_g1.json 2>/dev/null; cut -c1-330 gpurun_out/r3y_weak2_g1.json
