"""The `active` keyword through gpumd-mi (src/measure/active.cu: committee uncertainty over the `potential` lines of run.in, the
run follows the first; sigma_f = max_i sqrt(sum_d var_d) to active.out at every check step, the structure to active.xyz when it
exceeds the threshold).  Parsing on the CPU tier with --check-input, a run on the CPU tier through the emulator host and on the
GPU tier through gpumd-mi, both against the same quantity formed from the oracle's forces."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import helpers as H

EXE = os.path.join(H.ROOT, "gpumd_amd", "bin", "gpumd-mi")
EMU = os.path.join(H.ROOT, "tests", "emu", "gpumd-mi-emu")
NEP_A, NEP_B = H.golden("PbTe", "nep.txt"), H.golden("PbTe", "nep_B.txt")


@pytest.fixture(scope="module", autouse=True)
def _build():
    subprocess.run(["make", "-s", "-C", os.path.join(H.ROOT, "tests", "emu"), "all"], check=True)
    if not os.path.exists(EXE):
        import __graft_entry__ as g
        g.build()


def _run(exe, tmp_path, text, args=()):
    shutil.copy(H.golden("PbTe", "model.xyz"), tmp_path / "model.xyz")
    (tmp_path / "run.in").write_text(text)
    return subprocess.run([exe] + list(args), cwd=str(tmp_path), capture_output=True, text=True,
                          env=dict(os.environ, OMP_NUM_THREADS="1"))


def test_active_parse_and_errors(tmp_path):
    base = "potential %s\npotential %s\n" % (NEP_A, NEP_B)
    out = _run(EXE, tmp_path, base + "active 2 1 0 1 0.05\nrun 4\n", ["--check-input"])
    assert out.returncode == 0, out.stdout
    for line in ("Active learning.", "    check uncertainty every 2 steps.", "    with velocity data.", "    without force data.",
                 "    with per-atom uncertainty data.", "    will check if uncertainties exceed 0.050000 every 2 iterations."):
        assert line in out.stdout, line
    for args, msg in (("2 1 0 1", "active should have 5 parameters"), ("x 1 0 1 0.05", "check interval should be an integer"),
                      ("0 1 0 1 0.05", "check interval should > 0"), ("2 1 0 1 big", "threshold should be a real number")):
        out = _run(EXE, tmp_path, base + "active %s\nrun 4\n" % args, ["--check-input"])
        assert out.returncode == 1 and msg in out.stdout, out.stdout


def _check_run(exe, tmp_path):
    # two steps from rest on the 250-atom cell (small-box branch of both potentials); threshold 0: every check step is dumped
    text = ("potential %s\npotential %s\nvelocity 300 seed 3\nensemble nve\ntime_step 1\nactive 1 1 1 1 0.0\nrun 2\n" % (NEP_A, NEP_B))
    out = _run(exe, tmp_path, text)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = np.loadtxt(tmp_path / "active.out", ndmin=2)
    assert rows.shape == (2, 2) and np.allclose(rows[:, 0], [1.0, 2.0])
    frames = H.read_xyz_frames(str(tmp_path / "active.xyz"))
    assert len(frames) == 2
    for fr, (_, sigma) in zip(frames, rows):
        # the same uncertainty from the ORACLE's forces of the two models on the dumped coordinates
        fo = []
        for nep in (NEP_A, NEP_B):
            orc = H.Oracle(nep)
            typ = H.types_from_species(fr["species"], orc.symbols).astype(np.int32)
            _, f, _ = orc.compute(typ, fr["h"], H.soa(fr["pos"]), precision=64)
            fo.append(f.reshape(3, -1))
        fo = np.array(fo)
        var = (fo ** 2).mean(axis=0) - fo.mean(axis=0) ** 2
        unc = np.sqrt(var.sum(axis=0))
        assert abs(unc.max() - sigma) < 2e-4 * max(1.0, unc.max()), (unc.max(), sigma)
        # the per-atom column and the forces of the main potential (the first `potential` line)
        np.testing.assert_allclose(fr["uncertainty"].reshape(-1), unc, atol=3e-4)
        np.testing.assert_allclose(fr["forces"], fo[0].T, atol=3e-4)
        assert abs(float(fr["comment"]["uncertainty"]) - sigma) < 1e-7


def test_active_run_on_the_emulator_host(tmp_path):
    _check_run(EMU, tmp_path)


@pytest.mark.gpu
def test_active_run_on_the_gpu(tmp_path):
    _check_run(EXE, tmp_path)
