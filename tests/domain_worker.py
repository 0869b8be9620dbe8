"""Worker of tests/test_domain_gloo.py: one rank of a gloo process group running the domain
decomposition (gpumd_amd/domain.py) with the test-only kernel emulator as its engine."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from gpumd_amd.domain import DomainMD  # noqa: E402


def main():
    out_dir, reps, grid, nsteps, temp = sys.argv[1], eval(sys.argv[2]), eval(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
    on_gpu = len(sys.argv) > 6 and sys.argv[6] == "gpu"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # "gpu": the product library, all ranks sharing cuda:0 of a 1-GPU test box, messages staged
    # through host memory over gloo (RCCL refuses two ranks on one device)
    drv = H.GpuDriver() if on_gpu else H.EmuDriver()
    device = torch.device("cuda:0") if on_gpu else torch.device("cpu")
    nep = H.golden("PbTe", "nep.txt")
    model = drv.model(nep)
    h, typ, x = H.pbte_supercell(reps, rattle=0.02, seed=31)
    n = len(typ)
    mass = np.where(typ == 0, H.MASS["Te"], H.MASS["Pb"]).astype(np.float64)
    vel = H.maxwell_velocities(mass, temp, seed=5)
    mine = np.arange(n) % world == rank  # arbitrary initial distribution; setup() migrates
    X = torch.from_numpy(x.reshape(3, n)[:, mine].copy()).to(device)
    V = torch.from_numpy(vel.reshape(3, n)[:, mine].copy()).to(device)
    T = torch.from_numpy(typ[mine].copy()).to(device)
    M = torch.from_numpy(mass[mine].copy()).to(device)
    ids = torch.from_numpy(np.arange(n)[mine].astype(np.float64)).to(device)
    md = DomainMD(lambda cap: drv.engine(model, cap), model.info.rc_radial, h, (1, 1, 1), grid, rank, world,
                  device, stage_through_host=on_gpu)
    md.setup(X, V, T, M, ids=ids)
    md.initial_forces()
    i0, x0, v0, f0 = md.gather_owned()
    th0 = md.thermo()
    dt = 2.0 / H.TIME_UNIT
    md.run(nsteps, dt)
    i1, x1, v1, f1 = md.gather_owned()
    th1 = md.thermo()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), i0=i0, f0=f0, th0=th0, i1=i1, x1=x1, v1=v1, f1=f1, th1=th1,
             n_loc=md.n_loc, n_own=md.n_own, ndec=md.num_decompositions, nover=md.num_overlapped)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
