"""The C++ host (gpumd_amd/host) on the CPU tier: the same sources as gpumd-mi, linked against the kernel-logic emulator
library with the HIP runtime calls replaced by tests/emu/shim -- run.in / model.xyz parsing, the fused run segments, the
output files and the one-process-per-GPU mode (two ranks over the TCP transport) all run without a GPU."""
import os
import shutil
import socket
import subprocess

import numpy as np
import pytest

import helpers as H

EXE = os.path.join(H.ROOT, "tests", "emu", "gpumd-mi-emu")


@pytest.fixture(scope="module", autouse=True)
def _build():
    subprocess.run(["make", "-s", "-C", os.path.join(H.ROOT, "tests", "emu"), "all"], check=True)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _workdir(path, run_in):
    os.makedirs(path, exist_ok=True)
    shutil.copy(H.golden("PbTe", "model.xyz"), os.path.join(path, "model.xyz"))
    with open(os.path.join(path, "run.in"), "w") as f:
        f.write(run_in.replace("NEP", H.golden("PbTe", "nep.txt")))
    return path


def _run(wd, world=1):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NEPMI_TRANSPORT="tcp", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([EXE], cwd=wd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return outs[0]


def _thermo(wd):
    return np.loadtxt(os.path.join(wd, "thermo.out"))


RUN_IN = ("replicate 4 2 2\npotential NEP%s\nvelocity 1500 seed 7\nensemble %s\ntime_step 2\ndump_thermo 5\n"
          "dump_xyz 10 d.xyz velocity force\ndump_restart 20\nrun 20\n")


def test_single_rank_run_against_the_oracle(tmp_path):
    """gpumd-mi (fused segments between output steps) == the oracle's NVE loop on the same input."""
    wd = _workdir(str(tmp_path / "a"), RUN_IN % ("", "nve"))
    out = _run(wd)
    assert "Speed of this run" in out
    th = _thermo(wd)
    assert th.shape == (4, 18)
    # the same trajectory from the oracle: model.xyz replicated like the host does, its glibc rand() velocities read
    # back from the first dump is not possible -- compare with the host's own per-call path instead (next test) and
    # check physics here: energy conservation and the dump_xyz / restart files
    n = 250 * 16
    etot = th[:, 1] + th[:, 2]
    assert np.abs(etot - etot[0]).max() < 2e-3 * 4 * n
    fr = H.read_xyz_frames(os.path.join(wd, "d.xyz"))
    assert len(fr) == 2 and fr[0]["n"] == n and fr[0]["forces"].shape == (n, 3)
    rs = H.read_xyz_frames(os.path.join(wd, "restart.xyz"))[0]
    assert np.abs(rs["pos"] - fr[1]["pos"]).max() < 1e-4  # restart (full precision) vs the single-precision dump


@pytest.mark.parametrize("ens", ["nve", "nvt_ber 1500 1000 20", "nvt_nhc 1500 1000 20", "nvt_lan 1500 1000 20",
                                 "nvt_bao 1500 1500 20"])
def test_two_ranks_write_the_same_files_as_one(tmp_path, ens):
    """One process per GPU: thermo.out, dump_xyz (every atom gathered to rank 0 in file order) and restart.xyz of a
    2-rank run equal the single-rank run's to FP32 summation noise."""
    a = _workdir(str(tmp_path / "one"), RUN_IN % ("", ens))
    b = _workdir(str(tmp_path / "two"), RUN_IN % (" x", ens))
    _run(a, 1)
    out = _run(b, 2)
    assert "Use 2 GPUs: process grid 2 x 1 x 1" in out
    # 38 A slabs: the counted rule takes the reverse ghosts (shell rc + skin, forces returned to the owners), so the pressure
    # columns below also check the virial halves that stay on the ghosts
    assert "partial forces return to their owners" in out
    ta, tb = _thermo(a), _thermo(b)
    np.testing.assert_allclose(tb[:, :3], ta[:, :3], rtol=2e-6)
    np.testing.assert_allclose(tb[:, 3:9], ta[:, 3:9], rtol=1e-3, atol=1e-4)
    fa, fb = H.read_xyz_frames(os.path.join(a, "d.xyz")), H.read_xyz_frames(os.path.join(b, "d.xyz"))
    assert len(fa) == len(fb) == 2
    Hm = fa[0]["lattice"].T
    for x, y in zip(fa, fb):
        assert x["species"] == y["species"]
        d = np.linalg.solve(Hm, (x["pos"] - y["pos"]).T)
        d -= np.rint(d)
        assert np.abs(Hm @ d).max() < 1e-4
        assert np.abs(x["vel"] - y["vel"]).max() < 1e-5
        assert np.abs(x["forces"] - y["forces"]).max() < 1e-3
    ra, rb = H.read_xyz_frames(os.path.join(a, "restart.xyz"))[0], H.read_xyz_frames(os.path.join(b, "restart.xyz"))[0]
    assert np.abs(ra["vel"] - rb["vel"]).max() < 1e-6


def test_correct_velocity_and_direction_token(tmp_path):
    wd = _workdir(str(tmp_path / "c"), "replicate 2 2 2\npotential NEP z\nvelocity 300 seed 3\nensemble nve\ntime_step 1\n"
                                       "correct_velocity 10\ndump_thermo 10\ndump_xyz 20 d.xyz velocity mass\nrun 20\n")
    out = _run(wd)
    assert "Correct linear and angular momenta." in out and "every 10 steps." in out
    assert "the partition direction z is not used" in out
    fr = H.read_xyz_frames(os.path.join(wd, "d.xyz"))[0]
    p = (fr["mass"][:, 0:1] * fr["vel"]).sum(axis=0)
    assert np.abs(p).max() < 1e-3 * np.abs(fr["mass"][:, 0:1] * fr["vel"]).sum()  # momentum stays zero
    bad = _workdir(str(tmp_path / "d"), "potential NEP q\nrun 1\n")
    res = subprocess.run([EXE], cwd=bad, capture_output=True, text=True)
    assert res.returncode != 0 and "partition direction" in res.stdout


def test_nvt_lan_keyword_runs_and_thermostats(tmp_path):
    """`ensemble nvt_lan` (the thermostat of the reference's own examples/gpumd_dynamic/run.in): parsed, the generator seed
    drawn from the host's rand() stream like Ensemble_LAN does, a cold start is pulled towards the target."""
    run_in = "potential NEP\nvelocity 50\nensemble nvt_lan 600 600 5\ntime_step 1\ndump_thermo 10\nrun 40\n"
    wd = _workdir(str(tmp_path / "lan"), run_in)
    out = _run(wd)
    assert "choose the Langevin method." in out and "Langevin generator seed =" in out
    th = _thermo(wd)
    assert th.shape[0] == 4 and np.isfinite(th).all()
    assert th[-1, 0] > 250.0, th[:, 0]  # from 50 K towards 600 K with tau = 5 steps
