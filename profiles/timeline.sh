# device-timeline gaps between the kernels of the step: bash profiles/timeline.sh <tag> <bench args...>
T=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_$T -o bench -- python /root/repo/bench.py --no-cpu-baseline "$@" > /root/repo/gpurun_out/prof_$T.log 2>&1
cd /root/repo
python profiles/summarize_rocpd.py timeline $(ls gpurun_out/prof_$T/*.db | head -1) gpurun_out/${T}_timeline.csv
python profiles/summarize_rocpd.py stats $(ls gpurun_out/prof_$T/*.db | head -1) gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/prof_$T
cat gpurun_out/${T}_timeline.csv | cut -c1-220
tail -1 gpurun_out/prof_$T.log | cut -c1-300
