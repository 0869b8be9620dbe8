#!/bin/bash
# r3u: the interior/boundary split of the radial pass (nepmi_dist_set_overlap) on and off, in-process weak (2 ranks) and strong (8 ranks)
set -x
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
for ov in 0 1; do
  timeout 300 python profiles/inproc_weak.py --ranks 2 --steps 100 --overlap $ov > gpurun_out/r3u_weak2_ov$ov.json 2>/dev/null; cut -c1-330 gpurun_out/r3u_weak2_ov$ov.json
  for g in 0 1; do
    timeout 300 python profiles/inproc_weak.py --strong --ranks 8 --steps 60 --ghosts $g --overlap $ov > gpurun_out/r3u_strong8_g${g}_ov$ov.json 2>/dev/null; cut -c1-100 gpurun_out/r3u_strong8_g${g}_ov$ov.json; grep -o '"ms_per_step_ranks[^,]*, "ms_per_step_one[^,]*, "work_inflation[^,]*' gpurun_out/r3u_strong8_g${g}_ov$ov.json
  done
done
timeout 300 python profiles/inproc_weak.py --ranks 4 --steps 60 --overlap 0 > gpurun_out/r3u_weak4_ov0.json 2>/dev/null; cut -c1-330 gpurun_out/r3u_weak4_ov0.json
