# round 6s: the in-process weak-scaling proxy with the ranks' kernels SERIALISED (GPU_MAX_HW_QUEUES=1) under rocprofv3: the sum of the
# kernel durations of two ranks against the one-domain run of the same atoms -- the decomposition's real extra work, without the
# proxy's host-side barriers and without kernels of two ranks slowing each other down
export GPU_MAX_HW_QUEUES=1
cd /tmp && export TMPDIR=/tmp
for leg in ranks one; do
timeout 400 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_ip_$leg -o t -- python /root/repo/profiles/inproc_weak.py --ranks 2 --steps 100 --only $leg > /root/repo/gpurun_out/prof_ip_$leg.log 2>&1
python /root/repo/profiles/summarize_rocpd.py stats $(ls /root/repo/gpurun_out/prof_ip_$leg/*.db | head -1) /root/repo/gpurun_out/r6s_ip_${leg}_kernel_stats.csv
rm -rf /root/repo/gpurun_out/prof_ip_$leg
done
cd /root/repo
python - <<'PY'
import csv
def tot(f):
    rows=list(csv.DictReader(open(f)))
    step=[r for r in rows if any(k in r["kernel"] for k in ("RadialWin2","AngularFused","force_scatter","ForceFold","ResidentStep","Halo","Ghost","Vote","Pack","Unpack"))]
    return sum(float(r["total_us"]) for r in step), {r["kernel"][:60]:(int(r["calls"]),round(float(r["avg_us"]),1)) for r in step}
a,ka=tot("gpurun_out/r6s_ip_ranks_kernel_stats.csv"); b,kb=tot("gpurun_out/r6s_ip_one_kernel_stats.csv")
print("sum of step-kernel time: 2 ranks %.1f ms, one domain %.1f ms, ratio %.3f"%(a/1e3,b/1e3,a/b))
for k,v in ka.items(): print("ranks",k,v)
for k,v in kb.items(): print("one  ",k,v)
PY
