import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import helpers as H
drv = H.GpuDriver()
nep, (h, typ, x) = H.golden("PbTe", "nep.txt"), H.pbte_supercell((6, 6, 6), rattle=0.03, seed=17)
mass = np.where(typ == 0, H.MASS["Te"], H.MASS["Pb"]).astype(np.float64)
n = len(typ)
vel = H.maxwell_velocities(mass, 2500.0, seed=4)
m = drv.model(nep)
def run(brick, mask, steps):
    eng = drv.engine(m, n)
    eng.set_win_lanes(1); eng.set_force_form(1); eng.set_brick_force(brick); eng.set_radial_mask(mask)
    d_t, d_m, d_x, d_v = drv.dev(typ), drv.dev(mass), drv.dev(x), drv.dev(vel)
    d_pe, d_f, d_w = drv.zeros(n), drv.zeros(3 * n), drv.zeros(9 * n)
    eng.force_compute(h, d_t, d_x, d_pe, d_f, d_w)
    f0 = drv.host(d_f).copy()
    th = eng.run_nve(h, d_t, d_m, 2.0 / H.TIME_UNIT, steps, d_x, d_v, d_pe, d_f, d_w, thermo_every=steps)
    return f0, drv.host(d_x), drv.host(d_f), eng.describe(), eng.stats().num_rebuild
for steps in (1, 2, 5, 12, 40):
    r = {}
    for key, (brick, mask) in {"brick": (True, False), "brick2": (True, False), "sep": (False, False), "mask": (False, True), "brickmask": (True, True)}.items():
        r[key] = run(brick, mask, steps)
    print(steps, "rebuilds", r["sep"][4], {k: (float(np.abs(v[0]-r["sep"][0]).max()), float(np.abs(v[1]-r["sep"][1]).max()), float(np.abs(v[2]-r["sep"][2]).max())) for k, v in r.items()})
print(r["brickmask"][3])
