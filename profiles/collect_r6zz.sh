# round 6, the last run on the final tree: the full profile set (collect_r6z.sh, files r6z_*), then the GPU tier
cd /root/repo
bash profiles/collect_r6z.sh > gpurun_out/r6z_collect.log 2>&1
tail -30 gpurun_out/r6z_collect.log | cut -c1-300
(timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/pytest_r6zz.log 2>&1; grep -E "passed|failed|^FAILED" gpurun_out/pytest_r6zz.log | head -5
