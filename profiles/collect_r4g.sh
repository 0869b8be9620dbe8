#!/bin/bash
# r4g: direct halo (one grouped exchange with the <= 26 grid neighbours) + scatter form in the decomposed driver: the dist tests on
# the GPU tier, then the in-process weak (2, 4 ranks) and strong (8 ranks, both ghost forms) measurements of r3u's protocol
set -x
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4g
timeout 1500 python -m pytest tests/test_dist.py tests/test_dist_inproc.py tests/test_bench_launch.py tests/test_host_cli.py -x -q -m gpu > gpurun_out/${T}_pytest_dist.log 2>&1
tail -5 gpurun_out/${T}_pytest_dist.log
timeout 300 python profiles/inproc_weak.py --ranks 2 --steps 100 > gpurun_out/${T}_weak2.json 2>/dev/null; cut -c1-330 gpurun_out/${T}_weak2.json
timeout 300 python profiles/inproc_weak.py --ranks 4 --steps 60 > gpurun_out/${T}_weak4.json 2>/dev/null; cut -c1-330 gpurun_out/${T}_weak4.json
for g in 0 1; do
  timeout 300 python profiles/inproc_weak.py --strong --ranks 8 --steps 60 --ghosts $g > gpurun_out/${T}_strong8_g${g}.json 2>/dev/null; cut -c1-100 gpurun_out/${T}_strong8_g${g}.json; grep -o '"ms_per_step_ranks[^,]*, "ms_per_step_one[^,]*, "work_inflation[^,]*' gpurun_out/${T}_strong8_g${g}.json
done
timeout 300 python profiles/inproc_weak.py --strong --ranks 2 --steps 60 > gpurun_out/${T}_strong2.json 2>/dev/null; grep -o '"ms_per_step_ranks[^,]*, "ms_per_step_one[^,]*, "work_inflation[^,]*' gpurun_out/${T}_strong2.json
