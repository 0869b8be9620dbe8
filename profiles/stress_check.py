"""Longer / odd-shaped runs of the product path that the parity tests are too short for.  Prints one line
per scenario; any exception or non-finite number is a failure.  python profiles/stress_check.py"""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import gpumd_amd  # noqa: E402
import helpers as H  # noqa: E402

dev = torch.device("cuda", 0)
dt = 1.0 / H.TIME_UNIT


def setup(nep, h, typ, x, mass, T0, seed=1, pbc=(1, 1, 1)):
    model = gpumd_amd.Model(nep)
    n = len(typ)
    vel = H.maxwell_velocities(mass, T0, seed=seed)
    eng = gpumd_amd.NEP(model, n, pbc=pbc)
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (typ, x, vel, mass)]
    pe, f, w = (torch.zeros(k * n, dtype=torch.float64, device=dev) for k in (1, 3, 9))
    eng.force_compute(h, t[0], t[1], pe, f, w)
    return eng, t, pe, f, w, n


def report(name, fn):
    try:
        print("%-34s %s" % (name, fn()), flush=True)
    except Exception as ex:  # noqa: BLE001
        print("%-34s FAILED: %s" % (name, str(ex).splitlines()[0][:160]), flush=True)
        traceback.print_exc(file=sys.stderr)


def nvt(kind):
    def run():
        from gpumd_amd import structures as S
        h, typ, x, mass, _ = S.pbte_block((10, 10, 10))
        eng, t, pe, f, w, n = setup(H.golden("PbTe", "nep.txt"), h, typ, x, mass, 300.0)
        fn = getattr(eng, "run_nvt_" + kind)
        th = fn(h, t[0], t[3], dt, 1500, 300.0, 300.0, 100.0, t[1], t[2], pe, f, w, thermo_every=500)
        assert np.isfinite(th).all()
        return "T = %s K, rebuilds %d" % (np.round(th[:, 0], 1).tolist(), eng.stats().num_rebuild)
    return run


def family(workload, reps, steps):
    def run():
        label, nep, h, typ, x, mass, vel = bench.build_workload(workload, reps, 7)
        eng, t, pe, f, w, n = setup(nep, h, typ, x, mass, 300.0)
        th = eng.run_nve(h, t[0], t[3], dt, steps, t[1], t[2], pe, f, w, thermo_every=steps // 2)
        e = (th[:, 1] + 1.5 * n * H.K_B * th[:, 0]) / n
        st = eng.stats(True)
        assert np.isfinite(th).all()
        return "%d atoms, %d steps: dE/N %.1e eV, T %.0f K, rebuilds %d, max nn %d/%d/%d, window mode %d" % (
            n, steps, abs(e[-1] - e[0]), th[-1, 0], st.num_rebuild, st.max_nn_skin, st.max_nn_radial, st.max_nn_angular,
            st.radial_tiles)
    return run


def slab():
    # periodic in x, y; free in z (vacuum): the window kernels see empty cells and open boundaries
    h, typ, x = H.rocksalt_orthogonal((12, 12, 6))
    n = len(typ)
    h = np.array(h, dtype=np.float64)
    h[8] *= 3.0
    mass = np.where(typ == 0, H.MASS["Te"], H.MASS["Pb"]).astype(np.float64)
    eng, t, pe, f, w, n = setup(H.golden("PbTe", "nep.txt"), h, typ, x, mass, 300.0, pbc=(1, 1, 0))
    th = eng.run_nve(h, t[0], t[3], dt, 600, t[1], t[2], pe, f, w, thermo_every=300)
    e = (th[:, 1] + 1.5 * n * H.K_B * th[:, 0]) / n
    assert np.isfinite(th).all()
    return "%d atoms, free z: dE/N %.1e eV, T %.0f K, window mode %d" % (n, abs(e[-1] - e[0]), th[-1, 0],
                                                                        eng.stats().radial_tiles)


def tiny():
    out = []
    for n_side in (1, 2):
        h, typ, x = H.rocksalt_orthogonal((n_side, n_side, n_side))  # 8 / 64 atoms: small-box branch
        n = len(typ)
        mass = np.where(typ == 0, H.MASS["Te"], H.MASS["Pb"]).astype(np.float64)
        eng, t, pe, f, w, n = setup(H.golden("PbTe", "nep.txt"), h, typ, x, mass, 300.0)
        th = eng.run_nve(h, t[0], t[3], dt, 200, t[1], t[2], pe, f, w, thermo_every=100)
        assert np.isfinite(th).all()
        out.append("%d atoms T %.0f K" % (n, th[-1, 0]))
    return ", ".join(out)


if __name__ == "__main__":
    for kind in ("ber", "nhc", "bdp"):
        report("PbTe 250k NVT " + kind + " 1500 steps", nvt(kind))
    report("carbon 1M NVE", family("carbon", (10, 10, 10), 300))
    report("UNEP-v1 256k NVE", family("unep", (10, 10, 10), 300))
    report("PbTe slab (pbc 1 1 0)", slab)
    report("PbTe tiny cells (small box)", tiny)
