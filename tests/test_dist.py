"""The N > 1 path: the C++ domain-decomposed driver of libnepmi (nepmi_dist_*: spatial decomposition, ghost exchange,
migration, the global skin vote, every ensemble) with world_size 2 and 4 over the TCP transport, against the
single-domain run of the same system.  CPU tier: kernel logic from the emulator library; GPU tier: the product
library with the ranks sharing the test box's one GPU (what is NOT covered is only the RCCL transport itself)."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import helpers as H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(world, spec):
    out = tempfile.mkdtemp(prefix="nepmi_dist_")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(H.ROOT, "tests", "dist_worker.py"), out, json.dumps(spec)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=900)[0].decode() for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    return [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(world)]


def _merge(ranks, n):
    ids = np.concatenate([r["i1"] for r in ranks])
    assert len(ids) == n and len(np.unique(ids)) == n  # every atom owned exactly once
    order = np.argsort(ids)
    x = np.concatenate([r["x1"] for r in ranks], axis=1)[:, order]
    v = np.concatenate([r["v1"] for r in ranks], axis=1)[:, order]
    f = np.concatenate([r["f1"] for r in ranks], axis=1)[:, order]
    ids0 = np.concatenate([r["i0"] for r in ranks])
    f0 = np.concatenate([r["f0"] for r in ranks], axis=1)[:, np.argsort(ids0)]
    return x, v, f, f0


def _check(world, spec, n, f_tol=3e-5, traj_tol=2e-6):
    multi = _run_ranks(world, spec)
    single = _run_ranks(1, dict(spec, grid=[1, 1, 1]))
    xs, vs, fs, f0s = _merge(single, n)
    xm, vm, fm, f0m = _merge(multi, n)
    # initial forces: the decomposed evaluation equals the single-domain one to FP32 summation order
    assert np.abs(f0m - f0s).max() < f_tol + 1e-5 * np.abs(f0s).max()
    # trajectory: positions modulo lattice vectors, velocities, forces
    assert np.abs(vm - vs).max() < traj_tol
    assert np.abs(fm - fs).max() < 30 * f_tol + 1e-4 * np.abs(fs).max()
    # global thermodynamic sums agree on every rank and with the single-domain run
    for r in multi:
        np.testing.assert_allclose(r["th1"], multi[0]["th1"], rtol=0, atol=0)
        np.testing.assert_allclose(r["th1"][:2], single[0]["th1"][:2], rtol=1e-6)
        if len(r["th"]):
            np.testing.assert_allclose(r["th"][:, :2], single[0]["th"][:, :2], rtol=1e-6)
    return multi, single


CASES = [
    # (world, model, reps, grid, ensemble, nsteps, temperature): hot enough for re-decompositions inside the run
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nve", 20, 3000.0),
    (4, "PbTe-reps", (3, 3, 2), (2, 2, 1), "nve", 20, 3000.0),
    (4, "PbTe-reps", (8, 2, 2), (4, 1, 1), "nve", 20, 3000.0),     # slabs: the grid of bench.py's weak scaling (N x 1 x 1)
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nvt_ber", 16, 2000.0),
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nvt_nhc", 16, 2000.0),
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nvt_bdp", 16, 2000.0),
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nvt_lan", 16, 2000.0),  # Langevin: an atom's noise does not depend on the decomposition
    (4, "PbTe-reps", (3, 3, 2), (2, 2, 1), "nvt_lan", 12, 2000.0),
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nvt_bao", 16, 2000.0),  # BAOAB
    # hot and long enough for a re-decomposition inside the Langevin loops: the generator states migrate with their atoms,
    # the loop resumes after the first half-step (resume_after_vv1 skips B-A-O-A), the kernels of the frozen steps do nothing
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nvt_lan", 24, 3000.0),
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nvt_bao", 24, 3000.0),
    (2, "C-2022", (12, 6, 6), (2, 1, 1), "nve", 10, 3000.0),       # config 5's model through the ghost levels
    (2, "C-2022", (12, 6, 6), (2, 1, 1), "nvt_ber", 10, 3000.0),   # config 5: NVT on the decomposed path
    (2, "UNEP-v1", (10, 5, 5), (2, 1, 1), "nve", 8, 3000.0),       # config 4's model (16 types, ZBL)
]


def _spec(device, model, reps, grid, ensemble, nsteps, temp, ghosts=None):
    return {"device": device, "model": model, "reps": list(reps), "grid": list(grid), "ensemble": ensemble, "nsteps": nsteps,
            "temp": temp, "dt_fs": 2.0, "t1": temp, "t2": 0.5 * temp, "tcoup": 20.0, "thermo_every": 4, "seed": 777,
            "ghosts": ghosts}


def _natoms(model, reps):
    per = {"PbTe-reps": 250, "C-2022": 8, "UNEP-v1": 4}[model]
    return per * reps[0] * reps[1] * reps[2]


@pytest.mark.parametrize("world,model,reps,grid,ensemble,nsteps,temp", CASES)
def test_decomposed_run_matches_single_domain(world, model, reps, grid, ensemble, nsteps, temp):
    multi, _ = _check(world, _spec("cpu", model, reps, grid, ensemble, nsteps, temp), _natoms(model, reps))
    if model == "PbTe-reps" and (ensemble == "nve" or (temp >= 3000.0 and nsteps >= 24)):
        assert max(int(r["ndec"]) for r in multi) >= 2  # atoms really moved past skin/2: migration + new ghosts


REVERSE_CASES = [
    # reverse-mode ghosts (nepmi_dist_set_ghost_mode 1): shell rc + skin, the ghosts' partial forces travel back to the owners
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nve", 20, 3000.0),
    (4, "PbTe-reps", (3, 3, 2), (2, 2, 1), "nve", 20, 3000.0),      # two stages: edge ghosts are forwarded, and so are their forces
    (8, "PbTe-reps", (3, 3, 3), (2, 2, 2), "nve", 12, 3000.0),      # three stages (the grid of a strong-scaling run on 8 GPUs)
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nvt_nhc", 16, 2000.0),
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nvt_lan", 16, 2000.0),
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nvt_bao", 16, 2000.0),
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nvt_bdp", 16, 2000.0),
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nvt_lan", 24, 3000.0),  # with a re-decomposition: the generator states migrate
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nvt_bao", 24, 3000.0),
    (2, "C-2022", (12, 6, 6), (2, 1, 1), "nve", 10, 3000.0),
    (2, "UNEP-v1", (10, 5, 5), (2, 1, 1), "nve", 8, 3000.0),        # ZBL: a pair potential, never computed on a ghost
]


def _check_reverse(device, world, model, reps, grid, ensemble, nsteps, temp):
    n = _natoms(model, reps)
    multi, single = _check(world, _spec(device, model, reps, grid, ensemble, nsteps, temp, ghosts=1), n)
    fwd = _run_ranks(world, _spec(device, model, reps, grid, "nve", 0, temp, ghosts=0))
    for r, q in zip(multi, fwd):
        assert int(r["reverse"]) == 1 and int(q["reverse"]) == 0
        assert int(r["n_own"]) <= int(r["n_loc"]) < int(q["n_loc"])  # the thinner shell
        # the six stress components: the virial halves left on the ghosts are part of the global sum
        np.testing.assert_allclose(r["th1"][2:], single[0]["th1"][2:], rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(r["th0"][2:], single[0]["th0"][2:], rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(r["th2"], r["th1"], rtol=1e-9, atol=1e-12)  # the same sums once the virials are home
    # per-atom virials in the order of the ids
    def virials(ranks):
        ids = np.concatenate([r["i1"] for r in ranks])
        return np.concatenate([r["w1"] for r in ranks], axis=1)[:, np.argsort(ids)]
    wm, ws = virials(multi), virials(single)
    assert np.abs(wm - ws).max() < 1e-4 + 2e-5 * np.abs(ws).max(), np.abs(wm - ws).max()
    return multi, single


@pytest.mark.parametrize("world,model,reps,grid,ensemble,nsteps,temp", REVERSE_CASES)
def test_reverse_ghosts_match_single_domain(world, model, reps, grid, ensemble, nsteps, temp):
    _check_reverse("cpu", world, model, reps, grid, ensemble, nsteps, temp)


@pytest.mark.parametrize("ghosts", [0, 1])
def test_open_boundary_in_the_decomposed_direction(ghosts):
    """A slab geometry: no periodic images along the decomposed direction, so the end ranks have one neighbour only (no
    message across the open faces, forward or reverse)."""
    spec = dict(_spec("cpu", "PbTe-reps", (6, 2, 2), (3, 1, 1), "nve", 16, 2000.0, ghosts=ghosts), pbc=[0, 1, 1])
    multi, _ = _check(3, spec, _natoms("PbTe-reps", (6, 2, 2)))
    assert all(int(r["reverse"]) == ghosts for r in multi)


def test_thin_subboxes_take_reverse_ghosts_by_the_counted_rule():
    """Five slabs of 15.2 A: thinner than the forward shell 2 (rc + skin) = 18 A, so the rule (no mode set) picks reverse
    ghosts; the lower and the upper shell of a slab overlap, i.e. an atom can be a ghost of both neighbours and its two
    returned force parts are added one after the other."""
    spec = _spec("cpu", "PbTe-reps", (4, 2, 2), (5, 1, 1), "nve", 16, 3000.0)
    multi, _ = _check(5, spec, _natoms("PbTe-reps", (4, 2, 2)))
    assert all(int(r["reverse"]) == 1 for r in multi)
    with pytest.raises(AssertionError, match="thinner than the ghost shell 2"):
        _run_ranks(5, dict(spec, ghosts=0, nsteps=0))


@pytest.mark.gpu
@pytest.mark.parametrize("world,model,reps,grid,ensemble,nsteps,temp", [
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nve", 20, 3000.0),
    (4, "PbTe-reps", (4, 4, 2), (2, 2, 1), "nve", 20, 3000.0),
    (8, "PbTe-reps", (4, 4, 4), (2, 2, 2), "nvt_nhc", 12, 3000.0),
    (2, "C-2022", (12, 6, 6), (2, 1, 1), "nve", 10, 3000.0),
    (2, "UNEP-v1", (10, 5, 5), (2, 1, 1), "nve", 8, 3000.0),
])
def test_reverse_ghosts_on_gpu_kernels(world, model, reps, grid, ensemble, nsteps, temp):
    _check_reverse("gpu", world, model, reps, grid, ensemble, nsteps, temp)


@pytest.mark.gpu
@pytest.mark.parametrize("world,model,reps,grid,ensemble,nsteps,temp,ghosts", [
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nve", 20, 3000.0, 0),
    (4, "PbTe-reps", (4, 4, 2), (2, 2, 1), "nve", 20, 3000.0, 0),
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nvt_nhc", 16, 2000.0, None),
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nvt_lan", 24, 3000.0, 0),  # with a re-decomposition: the generator states migrate
    (2, "C-2022", (12, 6, 6), (2, 1, 1), "nvt_ber", 10, 3000.0, None),
    (2, "UNEP-v1", (10, 5, 5), (2, 1, 1), "nve", 8, 3000.0, None),
    # the reverse-ghost form against the ORACLE as well (one and three exchange stages)
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nve", 20, 3000.0, 1),
    (8, "PbTe-reps", (4, 4, 4), (2, 2, 2), "nve", 12, 3000.0, 1),
    (2, "C-2022", (12, 6, 6), (2, 1, 1), "nve", 10, 3000.0, 1),
])
def test_decomposed_run_on_gpu_kernels(world, model, reps, grid, ensemble, nsteps, temp, ghosts):
    n = _natoms(model, reps)
    multi, _ = _check(world, _spec("gpu", model, reps, grid, ensemble, nsteps, temp, ghosts=ghosts), n)
    if ghosts is not None:
        assert all(int(r["reverse"]) == ghosts for r in multi)  # the form that was asked for is the one that ran
    # ... and against the ORACLE, not only against the single-domain run of the same library: the forces of the initial
    # configuration, and the forces the decomposed run ends with (after its re-decompositions, migrations and ghost
    # levels) at the positions it ends with
    xm, vm, fm, f0m = _merge(multi, n)
    if model == "PbTe-reps":
        nep, (h, typ, x) = H.golden("PbTe", "nep.txt"), H.pbte_supercell(tuple(reps), rattle=0.02, seed=31)
    elif model == "C-2022":
        nep, (h, typ, x) = H.golden("C", "nep.txt"), H.diamond(tuple(reps), 3.57, rattle=0.02, seed=32)
    else:
        nep, (h, typ, x) = H.golden("UNEP", "nep.txt"), H.fcc_alloy(tuple(reps), 3.9, 16, rattle=0.02, seed=33)
    orc = H.Oracle(nep)
    _, f64, _ = orc.compute(typ, h, x, precision=64, path=0)
    assert np.all(np.abs(f0m.reshape(-1) - f64) <= 1e-4 * np.abs(f64) + 3e-5), np.abs(f0m.reshape(-1) - f64).max()
    x1 = H.oracle_apply_pbc(h, np.ascontiguousarray(xm).reshape(-1))
    _, f64e, _ = orc.compute(typ, h, x1, precision=64, path=0)
    assert np.all(np.abs(fm.reshape(-1) - f64e) <= 1e-4 * np.abs(f64e) + 3e-5), np.abs(fm.reshape(-1) - f64e).max()


def test_decomposed_nve_matches_the_fused_single_gpu_loop():
    """The decomposed driver against nepmi_run_nve of the plain engine (not just against itself on one rank)."""
    spec = _spec("cpu", "PbTe-reps", (4, 2, 2), (2, 1, 1), "nve", 12, 3000.0)
    multi = _run_ranks(2, spec)
    n = _natoms("PbTe-reps", (4, 2, 2))
    xm, vm, fm, f0m = _merge(multi, n)
    drv = H.EmuDriver()
    nep = H.golden("PbTe", "nep.txt")
    h, typ, x = H.pbte_supercell((4, 2, 2), rattle=0.02, seed=31)
    typ = typ.astype(np.int32)
    mass = np.where(typ == 0, H.MASS["Te"], H.MASS["Pb"]).astype(np.float64)
    vel = H.maxwell_velocities(mass, 3000.0, seed=5)
    eng = drv.engine(drv.model(nep), n)
    d_t, d_m, d_x, d_v = drv.dev(typ), drv.dev(mass), drv.dev(x), drv.dev(vel)
    d_pe, d_f, d_w = drv.zeros(n), drv.zeros(3 * n), drv.zeros(9 * n)
    eng.force_compute(h, d_t, d_x, d_pe, d_f, d_w)
    f0 = drv.host(d_f).reshape(3, n)
    assert np.abs(f0m - f0).max() < 3e-5 + 1e-5 * np.abs(f0).max()
    th = eng.run_nve(h, d_t, d_m, 2.0 / H.TIME_UNIT, 12, d_x, d_v, d_pe, d_f, d_w, thermo_every=4)
    assert np.abs(vm - drv.host(d_v).reshape(3, n)).max() < 2e-6
    H3 = np.asarray(h).reshape(3, 3)
    frac = np.linalg.solve(H3, xm - drv.host(d_x).reshape(3, n))
    frac -= np.rint(frac)
    assert np.abs(H3 @ frac).max() < 2e-6
    np.testing.assert_allclose(multi[0]["th"][:, :2], th[:, :2], rtol=1e-6)
    np.testing.assert_allclose(multi[0]["th"][:, 2:], th[:, 2:], rtol=1e-4, atol=1e-6)


def test_overlapped_exchange_is_bit_identical_to_the_plain_order():
    """Interior bricks' radial pass BEFORE this step's ghosts arrive + boundary bricks after == the plain
    exchange-then-compute order, bit for bit; and the split path is really taken."""
    spec = _spec("cpu", "PbTe-reps", (8, 2, 2), (2, 1, 1), "nve", 12, 3000.0)
    a = _run_ranks(2, dict(spec, overlap=True))
    b = _run_ranks(2, dict(spec, overlap=False))
    assert all(int(r["nover"]) >= 6 for r in a) and all(int(r["nover"]) == 0 for r in b)
    for ra, rb in zip(a, b):
        assert np.array_equal(ra["i1"], rb["i1"])
        assert np.array_equal(ra["x1"], rb["x1"]) and np.array_equal(ra["v1"], rb["v1"]) and np.array_equal(ra["f1"], rb["f1"])


@pytest.mark.gpu
def test_rccl_transport_on_one_rank():
    """What a one-GPU box can exercise of the RCCL transport (two ranks on one device are refused by RCCL as a duplicate
    GPU): communicator set-up from the unique id, grouped send/recv (to the own rank, two messages to the same peer
    matched in order), the in-place all-reduces of every dtype the driver uses, and a decomposed run over it against
    the plain engine."""
    import ctypes as C
    import torch
    import gpumd_amd
    from gpumd_amd import _capi
    from gpumd_amd.dist import DistMD, Transport

    lib = gpumd_amd.load_library()
    dev = torch.device("cuda", 0)
    tr = Transport.rccl(lib, 0, 1, lambda ident: ident)
    t = tr.struct
    assert t.device_buffers == 1 and t.nranks == 1
    a = torch.arange(4096, dtype=torch.float64, device=dev)
    b = torch.zeros(4096, dtype=torch.float64, device=dev)
    sends = (_capi.NepmiMsg * 2)(_capi.NepmiMsg(a.data_ptr(), 8 * 1000, 0), _capi.NepmiMsg(a.data_ptr() + 8 * 1000, 8 * 3096, 0))
    recvs = (_capi.NepmiMsg * 2)(_capi.NepmiMsg(b.data_ptr(), 8 * 1000, 0), _capi.NepmiMsg(b.data_ptr() + 8 * 1000, 8 * 3096, 0))
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert t.exchange(t.ctx, 2, sends, 2, recvs, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    d = torch.tensor([1.5, -2.0], dtype=torch.float64, device=dev)
    i = torch.tensor([3, 7, -1], dtype=torch.int32, device=dev)
    l = torch.tensor([1 << 40], dtype=torch.int64, device=dev)
    assert t.allreduce(t.ctx, d.data_ptr(), 2, 0, 0, stream) == 0
    assert t.allreduce(t.ctx, i.data_ptr(), 3, 1, 1, stream) == 0
    assert t.allreduce(t.ctx, l.data_ptr(), 1, 2, 0, stream) == 0
    torch.cuda.synchronize()
    assert d.tolist() == [1.5, -2.0] and i.tolist() == [3, 7, -1] and l.tolist() == [1 << 40]

    # the skin vote posted inside the group of the next exchange (NEPMI_DT_DEFER; nepmi_transport_rccl with NEPMI_RCCL_FUSE_VOTE=1):
    # a collective and point-to-point operations in one ncclGroup -- what one rank can show is that the call sequence completes
    os.environ["NEPMI_RCCL_FUSE_VOTE"] = "1"
    try:
        tr2 = Transport.rccl(lib, 0, 1, lambda ident: ident)
    finally:
        del os.environ["NEPMI_RCCL_FUSE_VOTE"]
    t2 = tr2.struct
    assert t2.device_buffers == 3
    i2 = torch.tensor([5, -3], dtype=torch.int32, device=dev)
    b.zero_()
    assert t2.allreduce(t2.ctx, i2.data_ptr(), 2, 1 | 0x100, 1, stream) == 0   # deferred
    assert t2.exchange(t2.ctx, 2, sends, 2, recvs, stream) == 0                 # ... into this group
    assert t2.allreduce(t2.ctx, i2.data_ptr(), 2, 1 | 0x100, 1, stream) == 0   # deferred again
    assert t2.allreduce(t2.ctx, d.data_ptr(), 2, 0, 0, stream) == 0            # an ordinary one flushes it first
    torch.cuda.synchronize()
    assert torch.equal(a, b) and i2.tolist() == [5, -3] and d.tolist() == [1.5, -2.0]
    tr2.close()

    # a run of the decomposed driver over this transport == the plain fused loop
    drv = H.GpuDriver()
    nep = H.golden("PbTe", "nep.txt")
    h, typ, x = H.pbte_supercell((4, 4, 4), rattle=0.02, seed=31)
    typ = typ.astype(np.int32)
    n = len(typ)
    mass = np.where(typ == 0, H.MASS["Te"], H.MASS["Pb"]).astype(np.float64)
    vel = H.maxwell_velocities(mass, 1000.0, seed=5)
    model = drv.model(nep)
    md = DistMD(model, tr, h, (1, 1, 1), (1, 1, 1))
    md.setup(torch.from_numpy(typ).to(dev), torch.from_numpy(mass).to(dev), torch.from_numpy(x).to(dev),
             torch.from_numpy(vel).to(dev))
    md.compute()
    th_d = md.run("nve", 1.0 / H.TIME_UNIT, 40, 0.0, 0.0, 100.0, thermo_every=10)
    md.close()
    tr.close()
    eng = drv.engine(model, n)
    d_t, d_m, d_x, d_v = drv.dev(typ), drv.dev(mass), drv.dev(x), drv.dev(vel)
    d_pe, d_f, d_w = drv.zeros(n), drv.zeros(3 * n), drv.zeros(9 * n)
    eng.force_compute(h, d_t, d_x, d_pe, d_f, d_w)
    th = eng.run_nve(h, d_t, d_m, 1.0 / H.TIME_UNIT, 40, d_x, d_v, d_pe, d_f, d_w, thermo_every=10)
    np.testing.assert_allclose(np.asarray(th_d)[:, :2], th[:, :2], rtol=1e-7)
