# round 4: many-type lists ordered by type + run contraction: parity of the many-type cases, then the A/B on UNEP-v1
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size_parity.py tests/test_ref_md_parity.py -m gpu -x -q -k "UNEP or unep or BaZrO3 or many" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
export AB_ARGS="--no-extras --workload unep --steps 20 --warmup 5"
bash profiles/ab_run.sh 2>&1 | tee gpurun_out/r4ae_ab_unep_runs.txt
