# round 4, after the shared-table radial pass of many-type shapes: A/B on UNEP-v1, then the whole GPU tier
cd /root/repo
mkdir -p gpurun_out
export AB_ARGS="--no-extras --workload unep --steps 20 --warmup 5"
bash profiles/ab_run.sh 2>&1 | tee gpurun_out/r4aa_ab_unep_share.txt
python - <<'PY' 2>&1 | tee gpurun_out/r4aa_describe.txt
import json,subprocess,sys
for wl in ("pbte","unep","carbon"):
    out=subprocess.run([sys.executable,"bench.py","--no-cpu-baseline","--no-extras","--workload",wl,"--steps","4","--warmup","2"],capture_output=True,text=True).stdout.strip().splitlines()[-1]
    d=json.loads(out); print(wl, d["ms_per_step"], d["config"]["kernel_forms"])
PY
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4aa_pytest_gpu.log
cat gpurun_out/r4aa_pytest_gpu.log
