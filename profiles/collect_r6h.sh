cd /root/repo
export NEPMI_JIT=2
(timeout 600 python -m pytest tests/test_jit_shapes.py -m gpu -q -x) 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r6h -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-extras --workload carbon2024 --reps 10 10 10 --steps 10 --warmup 3 > /root/repo/gpurun_out/prof_r6h.log 2>&1
cd /root/repo
python profiles/summarize_rocpd.py stats $(ls gpurun_out/prof_r6h/*.db | head -1) gpurun_out/r6h_kernel_stats.csv
rm -rf gpurun_out/prof_r6h
head -8 gpurun_out/r6h_kernel_stats.csv | cut -c1-200
tail -1 gpurun_out/prof_r6h.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.4f' % d['ms_per_step'], d.get('thermo_last')[:3])"
