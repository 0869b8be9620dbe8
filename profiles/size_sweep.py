"""System-size sweep and NVE energy drift of the product path on one MI355X (PbTe, examples/nep_train/nep.txt).

  python profiles/size_sweep.py > profiles/r1_size_sweep.txt
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import gpumd_amd  # noqa: E402
import helpers as H  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    nep = H.golden("PbTe", "nep.txt")
    model = gpumd_amd.Model(nep)
    dt = 1.0 / H.TIME_UNIT
    print("# atoms  ms/step  atom-steps/s   (100 NVE steps after 10 warm-up, rebuilds included)")
    for reps in ((4, 4, 4), (8, 8, 8), (10, 10, 10), (16, 16, 16), (20, 20, 20), (25, 25, 25)):
        h, typ, x, mass, vel = bench.build_pbte(reps)
        n = len(typ)
        eng = gpumd_amd.NEP(model, n)
        t = [torch.from_numpy(a).to(dev) for a in (typ, x, vel, mass)]
        pe, f, w = (torch.zeros(k * n, dtype=torch.float64, device=dev) for k in (1, 3, 9))
        eng.force_compute(h, t[0], t[1], pe, f, w)
        eng.run_nve(h, t[0], t[3], dt, 10, t[1], t[2], pe, f, w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run_nve(h, t[0], t[3], dt, 100, t[1], t[2], pe, f, w)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print("%9d  %7.3f  %.3e" % (n, el / 100 * 1e3, n * 100 / el))
        if reps == (16, 16, 16):
            # energy conservation over 2000 further steps (1 fs): total energy per atom, drift and noise
            th = eng.run_nve(h, t[0], t[3], dt, 2000, t[1], t[2], pe, f, w, thermo_every=100)
            e = (th[:, 1] + 1.5 * n * H.K_B * th[:, 0]) / n
            print("#   NVE, 1,024,000 atoms, 2000 steps: E/N from %.9f to %.9f eV; max |dE/N| = %.2e eV, T = %.1f K"
                  % (e[0], e[-1], np.abs(e - e[0]).max(), th[-1, 0]))
        del eng, t, pe, f, w
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
