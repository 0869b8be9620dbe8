set -x
cd /root/repo
(timeout 900 python -m pytest tests/test_langevin.py tests/test_dist.py tests/test_dist_inproc.py tests/test_host_cli.py -m gpu -q -x) > gpurun_out/r3k_pytest.log 2>&1; grep -E "passed|failed|Error" gpurun_out/r3k_pytest.log | tail -5
