# round 6: what the host's look at the device flags costs at launch-bound sizes (NEPMI_POLL_EVERY: steps between two snapshots)
cd /root/repo
for pe in 4 16 64 100000; do
  for wl in "--workload si_tersoff --steps 2000 --warmup 200" "--reps 4 4 4 --steps 400 --warmup 40" "--reps 8 8 8 --steps 200 --warmup 20"; do
    NEPMI_POLL_EVERY=$pe timeout 120 python bench.py --no-cpu-baseline --no-extras $wl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('poll_every $pe', d['config']['workload'][:40], 'ms/step %.5f value %.4g' % (d['ms_per_step'], d['value']))"
  done
done
