cd /root/repo
export AB_ARGS="--no-extras --workload unep --reps 16 16 16 --steps 20 --warmup 5"
timeout 600 bash profiles/ab_run.sh
cp gpumd_amd/lib/variants/libnepmi_afpairs.so gpumd_amd/lib/libnepmi.so
(timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "UNEP") > gpurun_out/pytest_r6r.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_r6r.log | head -5
