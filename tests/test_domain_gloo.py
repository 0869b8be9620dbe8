"""CPU tier: the N > 1 path (spatial decomposition, ghost exchange, migration) with world_size 2
and 4 over gloo, kernel logic supplied by the emulator; results vs the single-domain run."""
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


def _run_ranks(world, reps, grid, nsteps, temp, device="cpu", overlap=True):
    out = tempfile.mkdtemp(prefix="nepmi_dom_")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1", NEPMI_OVERLAP="1" if overlap else "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(H.ROOT, "tests", "domain_worker.py"), out,
                                       repr(reps), repr(grid), str(nsteps), str(temp), device], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    return [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("world,reps,grid", [(2, (4, 2, 2), (2, 1, 1)), (4, (3, 3, 2), (2, 2, 1))])
def test_decomposed_run_matches_single_domain(world, reps, grid):
    _check_decomposed(world, reps, grid, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("world,reps,grid", [(2, (4, 2, 2), (2, 1, 1)), (4, (4, 4, 2), (2, 2, 1))])
def test_decomposed_run_on_gpu_kernels(world, reps, grid):
    """The same decomposition with the product library: ranks share the box's single GPU, payloads
    staged through the host over gloo (what is NOT covered here is only the RCCL transport)."""
    _check_decomposed(world, reps, grid, "gpu")


def test_overlapped_exchange_is_bit_identical_to_the_plain_order():
    """compute_levels_begin (interior bricks, BEFORE this step's ghosts arrive) + _end == the plain
    exchange-then-compute order, bit for bit; and the split path is really taken."""
    a = _run_ranks(2, (8, 2, 2), (2, 1, 1), 12, 3000.0, overlap=True)
    b = _run_ranks(2, (8, 2, 2), (2, 1, 1), 12, 3000.0, overlap=False)
    assert all(int(r["nover"]) >= 6 for r in a) and all(int(r["nover"]) == 0 for r in b)
    for ra, rb in zip(a, b):
        assert np.array_equal(ra["i1"], rb["i1"])
        assert np.array_equal(ra["x1"], rb["x1"]) and np.array_equal(ra["v1"], rb["v1"])
        assert np.array_equal(ra["f1"], rb["f1"]) and np.array_equal(ra["th1"], rb["th1"])


def _check_decomposed(world, reps, grid, device):
    nsteps, temp = 16, 8000.0  # hot: atoms cross domain faces and trigger re-decompositions
    res = _run_ranks(world, reps, grid, nsteps, temp, device)
    # single-domain reference through the plugin surface (emulator on CPU, library on the GPU)
    drv = H.EmuDriver() if device == "cpu" else H.GpuDriver()
    nep = H.golden("PbTe", "nep.txt")
    h, typ, x = H.pbte_supercell(reps, rattle=0.02, seed=31)
    n = len(typ)
    mass = np.where(typ == 0, H.MASS["Te"], H.MASS["Pb"]).astype(np.float64)
    vel = H.maxwell_velocities(mass, temp, seed=5)
    eng = drv.engine(drv.model(nep), n)
    d_t, d_m, d_x, d_v = drv.dev(typ), drv.dev(mass), drv.dev(x), drv.dev(vel)
    d_pe, d_f, d_w = drv.zeros(n), drv.zeros(3 * n), drv.zeros(9 * n)
    eng.force_compute(h, d_t, d_x, d_pe, d_f, d_w)
    f_ref0 = drv.host(d_f).reshape(3, n)
    pe_ref0 = drv.host(d_pe).sum()
    th = eng.run_nve(h, d_t, d_m, 2.0 / H.TIME_UNIT, nsteps, d_x, d_v, d_pe, d_f, d_w, thermo_every=nsteps)
    x_ref, v_ref, f_ref = drv.host(d_x).reshape(3, n), drv.host(d_v).reshape(3, n), drv.host(d_f).reshape(3, n)

    ids0 = np.concatenate([r["i0"] for r in res])
    assert sorted(ids0.tolist()) == list(range(n)), "every atom owned exactly once"
    f0 = np.concatenate([r["f0"] for r in res], axis=1)[:, np.argsort(ids0)]
    assert np.abs(f0 - f_ref0).max() < 5e-5  # FP32 rounding differs with the local origin
    np.testing.assert_allclose(res[0]["th0"][1], pe_ref0, rtol=1e-6)
    for r in res:
        assert r["n_loc"] > r["n_own"] > 0
    assert all(int(r["ndec"]) >= 2 for r in res), "the hot run must re-decompose at least once"

    ids1 = np.concatenate([r["i1"] for r in res])
    assert sorted(ids1.tolist()) == list(range(n))
    order = np.argsort(ids1)
    x1 = np.concatenate([r["x1"] for r in res], axis=1)[:, order]
    v1 = np.concatenate([r["v1"] for r in res], axis=1)[:, order]
    f1 = np.concatenate([r["f1"] for r in res], axis=1)[:, order]
    H3 = np.asarray(h).reshape(3, 3)
    frac = np.linalg.solve(H3, x1 - x_ref)
    frac -= np.rint(frac)
    assert np.abs(H3 @ frac).max() < 2e-6
    assert np.abs(v1 - v_ref).max() < 2e-6
    assert np.abs(f1 - f_ref).max() < 1e-4
    np.testing.assert_allclose(res[0]["th1"][:2], th[-1][:2], rtol=1e-5)
    np.testing.assert_allclose(res[0]["th1"][2:], th[-1][2:], rtol=1e-3, atol=1e-6)
