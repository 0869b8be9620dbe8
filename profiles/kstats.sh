# rocprofv3 kernel stats of one bench command: bash profiles/kstats.sh <tag> <bench args...>
T=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_$T -o bench -- python /root/repo/bench.py --no-cpu-baseline "$@" > /root/repo/gpurun_out/prof_$T.log 2>&1
cd /root/repo
python profiles/summarize_rocpd.py stats $(ls gpurun_out/prof_$T/*.db | head -1) gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/prof_$T
head -14 gpurun_out/${T}_kernel_stats.csv | cut -c1-200
