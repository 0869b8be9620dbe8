# round 6g: many-type radial pass, four candidates side by side without a branch (NEPMI_RW2_WIDE) -- same-box A/B on UNEP-v1, parity first
cd /root/repo
(timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "UNEP or BaZrO3 or cover") > gpurun_out/pytest_r6g.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_r6g.log | head
export AB_ARGS="--no-extras --workload unep --reps 16 16 16 --steps 20 --warmup 5"
timeout 600 bash profiles/ab_run.sh
