cd /tmp && export TMPDIR=/tmp
for leg in ranks one; do
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_ip_$leg -o t -- python /root/repo/profiles/inproc_weak.py --ranks 2 --steps 100 --only $leg > /root/repo/gpurun_out/prof_ip_$leg.log 2>&1
python /root/repo/profiles/summarize_rocpd.py stats $(ls /root/repo/gpurun_out/prof_ip_$leg/*.db | head -1) /root/repo/gpurun_out/ip_${leg}_kernel_stats.csv
rm -rf /root/repo/gpurun_out/prof_ip_$leg
echo "== $leg"; head -16 /root/repo/gpurun_out/ip_${leg}_kernel_stats.csv | cut -c1-120
done
