import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],3), {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()})
