cd /root/repo
export AB_ARGS="--no-extras --steps 40 --warmup 10"
bash profiles/ab_run.sh
export AB_ARGS="--no-extras --workload carbon --reps 10 10 10 --steps 30 --warmup 5"
bash profiles/ab_run.sh
export AB_ARGS="--no-extras --workload unep --reps 16 16 16 --steps 20 --warmup 5"
bash profiles/ab_run.sh
