import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import gpumd_amd, bench
from gpumd_amd import structures as H
label, nep_txt, h, typ, x, mass, vel = bench.build_workload("pbte", (6, 6, 6), 42)
n = len(typ); dev = torch.device("cuda", 0)
print("n", n, flush=True)
model = gpumd_amd.Model(nep_txt); print("model", flush=True)
eng = gpumd_amd.NEP(model, n); print("engine", flush=True)
t_type, t_mass = torch.from_numpy(typ).to(dev), torch.from_numpy(mass).to(dev)
t_x, t_v = torch.from_numpy(x).to(dev), torch.from_numpy(vel).to(dev)
t_pe, t_f, t_w = (torch.zeros(k * n, dtype=torch.float64, device=dev) for k in (1, 3, 9))
eng.force_compute(h, t_type, t_x, t_pe, t_f, t_w); torch.cuda.synchronize(); print("force", flush=True)
dt = 1.0 / H.TIME_UNIT
eng.run_nve(h, t_type, t_mass, dt, 5, t_x, t_v, t_pe, t_f, t_w); torch.cuda.synchronize(); print("warm", flush=True)
eng.set_timing(1); print("t1", flush=True)
eng.run_nve(h, t_type, t_mass, dt, 4, t_x, t_v, t_pe, t_f, t_w, thermo_every=4); print("probe", flush=True)
st = eng.stats(with_lists=False); print("stats", list(st.launches)[:8], flush=True)
eng.set_timing(0)
eng.set_timing(18); print("t18", flush=True)
eng.run_nve(h, t_type, t_mass, dt, 4, t_x, t_v, t_pe, t_f, t_w, thermo_every=4); torch.cuda.synchronize(); print("timed", flush=True)
st = eng.stats(with_lists=True); print("stats2", list(st.ms_kernel_sum)[:8], flush=True)
print(bench.gpu_state())
