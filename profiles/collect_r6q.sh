# round 6q: fused angular kernel, the workgroup's atoms sorted by their number of angular neighbours (NEPMI_AFU_SORT) -- parity, then same-box A/B
cd /root/repo
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size_parity.py -m gpu -q -x -k "not UNEP") > gpurun_out/pytest_r6q.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_r6q.log | head
export AB_ARGS="--no-extras --steps 40 --warmup 10"
timeout 600 bash profiles/ab_run.sh
export AB_ARGS="--no-extras --workload carbon --reps 10 10 10 --steps 30 --warmup 5"
timeout 600 bash profiles/ab_run.sh
