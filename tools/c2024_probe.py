# which kernels serve C_2024_NEP4 in diamond at a given size, and what they return (A/B: NEPMI_WIN_MAX_ATOMS)
#   python tools/c2024_probe.py CELLS OUT.npz [emu]      python tools/c2024_probe.py --compare A.npz B.npz
import os, sys, numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in ("pe", "f", "v"):
        d = np.abs(a[k] - b[k]).max()
        print(k, "max |a - b| = %.3e  (max |a| = %.3e)" % (d, np.abs(a[k]).max()))
    print("total energy: %.9f vs %.9f" % (a["pe"].sum(), b["pe"].sum()))
    sys.exit(0)
import helpers as H
drv = H.GpuDriver() if len(sys.argv) < 4 else H.EmuDriver()
pot = os.path.join(H.ROOT, "tests", "golden", "C", "nep_2024.txt")
c = int(sys.argv[1])
h, typ, x = H.diamond((c, c, c), 3.567, rattle=0.05, seed=3)
n = len(typ)
model = drv.model(pot)
eng = drv.engine(model, n)
xw, pe, f, v = H.engine_force(drv, eng, h, typ, x)
print(n, eng.describe())
np.savez(sys.argv[2], pe=pe, f=f, v=v)
