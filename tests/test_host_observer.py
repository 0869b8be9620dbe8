"""Several `potential` lines and dump_observer through gpumd-mi: Force's "observe" / "average" modes
(src/force/force.cu:55-73,213-217,514-565) and Dump_Observer (src/measure/dump_observer.cu:82-442)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import helpers as H

EXE = os.path.join(H.ROOT, "gpumd_amd", "bin", "gpumd-mi")
NEP_A, NEP_B = H.golden("PbTe", "nep.txt"), H.golden("PbTe", "nep_B.txt")


@pytest.fixture(scope="module", autouse=True)
def _build():
    if not os.path.exists(EXE):
        import __graft_entry__ as g
        g.build()


def _run(tmp_path, text, model=("PbTe", "model.xyz"), check_only=False):
    shutil.copy(H.golden(*model), tmp_path / "model.xyz")
    (tmp_path / "run.in").write_text(text)
    return subprocess.run([EXE] + (["--check-input"] if check_only else []), cwd=str(tmp_path), capture_output=True,
                          text=True)


def test_dump_observer_parse_and_errors(tmp_path):
    base = "potential %s\npotential %s\n" % (NEP_A, NEP_B)
    out = _run(tmp_path, base + "dump_observer observe 2 4 1 0\nrun 4\n", check_only=True)
    assert out.returncode == 0, out.stdout
    for line in ("Dump observer.", "    .out every 2 steps.", "    .exyz every 4 steps.", "    with velocity data.",
                 "    without force data.", "    evaluate all potentials, dumping .out every 2 and .exyz every 4 steps."):
        assert line in out.stdout, line
    for args, msg in (("observe 2 4 1", "dump_observer should have 5 parameters"),
                      ("watch 2 4 1 0", "observer mode should be 'observe' or 'average'"),
                      ("observe x 4 1 0", "dump interval thermo should be an integer"),
                      ("average 2 0 1 0", "dump interval exyz should > 0")):
        out = _run(tmp_path, base + "dump_observer %s\nrun 4\n" % args, check_only=True)
        assert out.returncode == 1 and msg in out.stdout, out.stdout


def test_inconsistent_or_non_nep_potentials_are_rejected(tmp_path):
    out = _run(tmp_path, "potential %s\npotential %s\nrun 1\n" % (NEP_A, H.golden("BaZrO3", "nep.txt")), check_only=True)
    assert out.returncode == 1 and "not consistent between the multiple potentials" in out.stdout
    # a non-NEP potential next to a NEP one (force.cu:213-217), either order, on an 8-atom silicon cell
    si, ters = H.golden("Si", "nep_3body.txt"), H.golden("Si", "Si_Tersoff_1989.txt")
    base = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0], [.25, .25, .25], [.25, .75, .75], [.75, .25, .75],
                     [.75, .75, .25]]) * 5.43
    cell = "8\nLattice=\"5.43 0 0 0 5.43 0 0 0 5.43\" Properties=species:S:1:pos:R:3\n" + "".join(
        "Si %.6f %.6f %.6f\n" % tuple(r) for r in base)
    for first, second in ((si, ters), (ters, si)):
        (tmp_path / "model.xyz").write_text(cell)
        (tmp_path / "run.in").write_text("potential %s\npotential %s\nrun 1\n" % (first, second))
        out = subprocess.run([EXE, "--check-input"], cwd=str(tmp_path), capture_output=True, text=True)
        assert out.returncode == 1 and "Multiple potentials may only be used with NEP potentials" in out.stdout, out.stdout


def _oracle_on_frame(nep, frame):
    orc = H.Oracle(nep)
    typ = H.types_from_species(frame["species"], orc.symbols)
    pe, f, v = orc.compute(typ.astype(np.int32), frame["h"], H.soa(frame["pos"]), precision=64, path=0)
    return pe.sum(), f.reshape(3, -1).T


@pytest.mark.gpu
def test_observe_mode_evaluates_every_potential_and_keeps_the_main_one_driving(tmp_path):
    text = ("replicate 2 2 2\npotential %s\npotential %s\nvelocity 300 seed 11\nensemble nve\ntime_step 1\ndump_thermo 2\n"
            "dump_xyz 2 main.xyz force precision double\ndump_observer observe 2 2 1 1\nrun 4\n" % (NEP_A, NEP_B))
    out = _run(tmp_path, text)
    assert out.returncode == 0, out.stdout + out.stderr
    thermo = np.loadtxt(tmp_path / "thermo.out")
    obs0, obs1 = np.loadtxt(tmp_path / "observer0.out"), np.loadtxt(tmp_path / "observer1.out")
    assert thermo.shape == obs0.shape == obs1.shape == (2, 18)
    # the main potential, evaluated again (while the engine is still timing its two force-assembly variants the second
    # evaluation may take the other one: FP32 summation order, not bits)
    np.testing.assert_allclose(obs0, thermo, rtol=2e-6, atol=1e-7)
    assert np.abs(obs1[:, 2] - obs0[:, 2]).min() > 1e-3              # the other model really is another model
    np.testing.assert_array_equal(obs1[:, 0], obs0[:, 0])           # same velocities: same temperature
    main = H.read_xyz_frames(str(tmp_path / "main.xyz"))
    f0, f1 = H.read_xyz_frames(str(tmp_path / "observer0.xyz")), H.read_xyz_frames(str(tmp_path / "observer1.xyz"))
    assert len(main) == len(f0) == len(f1) == 2 and f1[0]["n"] == 2000
    assert f1[0]["comment"]["properties"] == "species:S:1:pos:R:3:vel:R:3:forces:R:3"
    for k in range(2):
        # the run continues with the main potential's forces although the observer overwrote the arrays in between
        np.testing.assert_allclose(f0[k]["forces"], main[k]["forces"], atol=2e-6)
        e_b, f_b = _oracle_on_frame(NEP_B, main[k])
        np.testing.assert_allclose(float(f1[k]["comment"]["energy"]), e_b, rtol=1e-5)
        np.testing.assert_allclose(f1[k]["forces"], f_b, rtol=1e-4, atol=3e-5)
        np.testing.assert_allclose(obs1[k, 2], e_b, rtol=1e-5)
    # the trajectory is the single-potential one
    sub = tmp_path / "single"
    sub.mkdir()
    out = _run(sub, text.replace("potential %s\n" % NEP_B, "").replace("dump_observer observe 2 2 1 1\n", ""))
    assert out.returncode == 0, out.stdout
    np.testing.assert_allclose(np.loadtxt(sub / "thermo.out"), thermo, rtol=2e-6, atol=1e-7)


@pytest.mark.gpu
def test_average_mode_runs_on_the_mean_of_the_potentials(tmp_path):
    text = ("replicate 2 2 2\npotential %s\npotential %s\nvelocity 300 seed 11\nensemble nve\ntime_step 1\ndump_thermo 2\n"
            "dump_xyz 2 main.xyz force potential precision double\ndump_observer average 2 2 0 1\nrun 4\n" % (NEP_A, NEP_B))
    out = _run(tmp_path, text)
    assert out.returncode == 0, out.stdout + out.stderr
    assert not os.path.exists(tmp_path / "observer0.out")
    thermo, obs = np.loadtxt(tmp_path / "thermo.out"), np.loadtxt(tmp_path / "observer.out")
    np.testing.assert_allclose(obs, thermo, rtol=1e-9, atol=1e-12)  # the very same arrays, written twice
    main, fo = H.read_xyz_frames(str(tmp_path / "main.xyz")), H.read_xyz_frames(str(tmp_path / "observer.xyz"))
    assert fo[0]["comment"]["properties"] == "species:S:1:pos:R:3:forces:R:3"
    for k in range(2):
        (e_a, f_a), (e_b, f_b) = _oracle_on_frame(NEP_A, main[k]), _oracle_on_frame(NEP_B, main[k])
        np.testing.assert_allclose(main[k]["energy"], 0.5 * (e_a + e_b), rtol=1e-5)
        np.testing.assert_allclose(main[k]["forces"], 0.5 * (f_a + f_b), rtol=1e-4, atol=3e-5)
        np.testing.assert_allclose(fo[k]["forces"], main[k]["forces"], atol=1e-8)
        np.testing.assert_allclose(thermo[k, 2], 0.5 * (e_a + e_b), rtol=1e-5)
