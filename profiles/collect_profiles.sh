set -x
cd /root/repo
(timeout 900 python -m pytest tests -m gpu -q) > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|^E |^FAILED" gpurun_out/pytest_gpu.log | head
python bench.py > gpurun_out/bench_r1q.json 2> gpurun_out/bench_r1q.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r1q -o bench -- python /root/repo/bench.py --no-cpu-baseline > /root/repo/gpurun_out/prof_r1q.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d /root/repo/gpurun_out/prof_r1q_fetch -o bench -- python /root/repo/bench.py --no-cpu-baseline --steps 5 --warmup 2 > /root/repo/gpurun_out/prof_r1q_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /root/repo/gpurun_out/prof_r1q_write -o bench -- python /root/repo/bench.py --no-cpu-baseline --steps 5 --warmup 2 > /root/repo/gpurun_out/prof_r1q_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES -d /root/repo/gpurun_out/prof_r1q_sq -o bench -- python /root/repo/bench.py --no-cpu-baseline --steps 5 --warmup 2 > /root/repo/gpurun_out/prof_r1q_sq.log 2>&1
cd /root/repo
python profiles/summarize_rocpd.py stats $(ls gpurun_out/prof_r1q/*.db | head -1) gpurun_out/r1q_kernel_stats.csv
python profiles/summarize_rocpd.py pmc $(ls gpurun_out/prof_r1q_fetch/*.db | head -1) gpurun_out/r1q_pmc_fetch.csv
python profiles/summarize_rocpd.py pmc $(ls gpurun_out/prof_r1q_write/*.db | head -1) gpurun_out/r1q_pmc_write.csv
python profiles/summarize_rocpd.py pmc $(ls gpurun_out/prof_r1q_sq/*.db | head -1) gpurun_out/r1q_pmc_sq.csv
head -8 gpurun_out/r1q_kernel_stats.csv
tail -c 400 gpurun_out/bench_r1q.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
