"""gpumd-mi, the C++ host (gpumd_amd/host): run.in / model.xyz parsing on the CPU tier, a full
single-point + NVE run through the Force::compute surface on the GPU tier."""
import os
import subprocess

import numpy as np
import pytest

import helpers as H

EXE = os.path.join(H.ROOT, "gpumd_amd", "bin", "gpumd-mi")


def _workdir(tmp_path, run_in):
    import shutil
    shutil.copy(H.golden("PbTe", "model.xyz"), tmp_path / "model.xyz")
    (tmp_path / "run.in").write_text(run_in.replace("NEP", H.golden("PbTe", "nep.txt")))
    return str(tmp_path)


def _grouped_workdir(tmp_path, run_in, shift=0.0):
    """PbTe model.xyz with a charge column and two grouping methods (by species; by index parity + a third group)."""
    src = H.read_xyz_frames(H.golden("PbTe", "model.xyz"))[0]
    lines = open(H.golden("PbTe", "model.xyz")).read().split("\n")
    head = lines[1].split("Properties=")[0]
    out = ["%d" % src["n"], head + "Properties=species:S:1:pos:R:3:charge:R:1:group:I:2"]
    for k in range(src["n"]):
        sp = src["species"][k]
        x = src["pos"][k] + shift  # shift > 0 moves some atoms out of the cell: unwrapped != wrapped from step 0
        out.append("%s %.17g %.17g %.17g %g %d %d" % (sp, x[0], x[1], x[2], 0.5 if sp == "Pb" else -0.5,
                                                      0 if sp == "Te" else 1, 2 if k >= 240 else k % 2))
    (tmp_path / "model.xyz").write_text("\n".join(out) + "\n")
    (tmp_path / "run.in").write_text(run_in.replace("NEP", H.golden("PbTe", "nep.txt")))
    return str(tmp_path), src


@pytest.fixture(scope="module", autouse=True)
def _build():
    if not os.path.exists(EXE):
        import __graft_entry__ as g
        g.build()


def test_check_input_parses_run_in_and_model(tmp_path):
    wd = _workdir(tmp_path, "replicate 2 3 1   # comment\npotential NEP\nvelocity 300 seed 42\nensemble nve\n"
                            "time_step 2\ndump_thermo 5\ndump_xyz 10 d.xyz force velocity precision single\nrun 10\n")
    out = subprocess.run([EXE, "--check-input"], cwd=wd, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Number of atoms is 250." in out.stdout
    assert "Replicated the box: 2 x 3 x 1, 1500 atoms." in out.stdout
    assert "elements: Te Pb" in out.stdout
    assert "Time step for this run is 2 fs." in out.stdout
    assert "Run 10 steps." in out.stdout


def test_check_input_groups_and_dump_xyz_options(tmp_path):
    """read_xyz.cu:243-312 (charge, group:I:k), group.cu:25-72, replicate.cu:60-84, Dump_XYZ::parse."""
    wd, _ = _grouped_workdir(tmp_path, "replicate 1 1 2\npotential NEP\nensemble nve\ntime_step 1\n"
                                       "dump_xyz 5 frames/f_* group 1 2 precision single mass charge velocity force "
                                       "potential unwrapped_position virial group_labels\nrun 10\n")
    out = subprocess.run([EXE, "--check-input"], cwd=wd, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    o = out.stdout
    assert "Have 2 grouping method(s)." in o
    assert "There are 2 groups of atoms in grouping method 0." in o
    assert "There are 3 groups of atoms in grouping method 1." in o
    assert "    250 atoms in group 0." in o and "    20 atoms in group 2." in o  # after replicate 1 1 2
    for line in ("Dump extended XYZ.", "    every 5 steps.", "    into file frames/f_*.", "    with single precision.",
                 "    grouping method is 1 and group ID is 2.", "    has unwrapped position.", "    has group labels.",
                 "    has charge specified in model.xyz."):
        assert line in o, line
    assert "for the whole system" not in o


@pytest.mark.parametrize("args,msg", [
    ("0 2 5 d.xyz", "no longer takes <grouping_method> <group_id>"),
    ("x d.xyz", "dump interval should be an integer"),
    ("0 d.xyz", "dump interval should > 0"),
    ("5 d.xyz force force", "Quantity 'force' is specified more than once"),
    ("5 d.xyz group", "now called 'group_labels'"),
    ("5 d.xyz group 2 0", "Grouping method should < number of grouping methods"),
    ("5 d.xyz group 1 3", "Group ID should < number of groups"),
    ("5 d.xyz group 0 0 group 0 1", "Option 'group' is specified more than once"),
    ("5 d.xyz precision half", "Invalid precision"),
    ("5 d.xyz precision", "Not enough arguments for option 'precision'"),
    ("5 d.xyz bec", "Cannot output BEC"),
    ("5 d.xyz pressure", "Unrecognized argument in dump_xyz")])
def test_dump_xyz_input_errors(tmp_path, args, msg):
    wd, _ = _grouped_workdir(tmp_path, "potential NEP\ndump_xyz %s\nrun 1\n" % args)
    out = subprocess.run([EXE, "--check-input"], cwd=wd, capture_output=True, text=True)
    assert out.returncode == 1
    assert "Input Error" in out.stdout and msg in out.stdout, out.stdout


def test_group_labels_need_a_grouping_method(tmp_path):
    wd = _workdir(tmp_path, "potential NEP\ndump_xyz 5 d.xyz group_labels\nrun 1\n")
    out = subprocess.run([EXE, "--check-input"], cwd=wd, capture_output=True, text=True)
    assert out.returncode == 1 and "Cannot output group labels without a grouping method" in out.stdout


@pytest.mark.parametrize("bad,msg", [("potential NEP\nfoo 1\nrun 1\n", "invalid keyword"),
                                     ("potential NEP\nensemble nvt_qtb 300 300 100\nrun 1\n", "not available"),
                                     ("potential NEP\nensemble nvt_nhc 300 300 0.5\nrun 1\n", "coupling should >= 1"),
                                     ("velocity 300\nrun 1\n", "no 'potential'")])
def test_input_errors_exit_like_the_reference(tmp_path, bad, msg):
    wd = _workdir(tmp_path, bad)
    out = subprocess.run([EXE, "--check-input"], cwd=wd, capture_output=True, text=True)
    assert out.returncode == 1
    assert "Input Error" in out.stdout and msg in out.stdout


@pytest.mark.gpu
def test_single_point_matches_oracle(tmp_path):
    """examples/gpumd_static/run.in (time_step 0, dump force) on a 2x2x2 supercell."""
    wd = _workdir(tmp_path, "replicate 2 2 2\npotential NEP\nvelocity 1\nensemble nve\ntime_step 0\n"
                            "dump_xyz 1 dump.xyz force potential precision double\nrun 1\n")
    out = subprocess.run([EXE], cwd=wd, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "atom*step/second" in out.stdout
    fr = H.read_xyz_frames(os.path.join(wd, "dump.xyz"))[0]
    src = H.read_xyz_frames(H.golden("PbTe", "model.xyz"))[0]
    orc = H.Oracle(H.golden("PbTe", "nep.txt"))
    typ0 = H.types_from_species(src["species"], orc.symbols)
    h, typ, pos = H.replicate(src["h"], typ0, src["pos"], (2, 2, 2))
    x = H.oracle_apply_pbc(h, H.soa(pos))
    pe, f, v = orc.compute(typ.astype(np.int32), h, x, precision=64, path=0)
    assert fr["n"] == 2000
    np.testing.assert_allclose(fr["energy"], pe.sum(), rtol=1e-5)
    np.testing.assert_allclose(fr["forces"], f.reshape(3, -1).T, rtol=1e-4, atol=3e-5)
    np.testing.assert_allclose(fr["energy_atom"][:, 0], pe, rtol=1e-5, atol=2e-5)
    vt = v.reshape(9, -1).sum(axis=1)
    np.testing.assert_allclose(fr["virial"], [vt[0], vt[3], vt[4], vt[3], vt[1], vt[5], vt[4], vt[5], vt[2]],
                               rtol=1e-4, atol=5e-2)


@pytest.mark.gpu
def test_config1_gpumd_static_example(tmp_path):
    """BASELINE config 1 end to end: the reference's examples/gpumd_static (PbTe 250 atoms, small-box
    branch) run by gpumd-mi; dump.xyz against the dump.xyz the reference's CUDA build wrote."""
    run_in = open(H.golden("PbTe", "run.in")).read().replace("../nep_train/nep.txt", "NEP")
    wd = _workdir(tmp_path, run_in)
    out = subprocess.run([EXE], cwd=wd, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    got = H.read_xyz_frames(os.path.join(wd, "dump.xyz"))[0]
    ref = H.read_xyz_frames(H.golden("PbTe", "dump.xyz"))[0]
    np.testing.assert_allclose(got["energy"], ref["energy"], rtol=1e-6)
    assert np.abs(got["forces"] - ref["forces"]).max() < 2e-5       # check_force.m:9
    np.testing.assert_allclose(got["virial"], ref["virial"], rtol=1e-4, atol=2e-3)
    assert np.abs(got["pos"] - ref["pos"]).max() < 1e-12


@pytest.mark.gpu
def test_nve_run_writes_thermo(tmp_path):
    wd = _workdir(tmp_path, "replicate 2 2 2\npotential NEP\nvelocity 300 seed 42\nensemble nve\ntime_step 1\n"
                            "dump_thermo 10\ndump_restart 40\nrun 40\n")
    out = subprocess.run([EXE], cwd=wd, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    th = np.loadtxt(os.path.join(wd, "thermo.out"))
    assert th.shape == (4, 18)
    etot = th[:, 1] + th[:, 2]
    assert np.abs(etot - etot[0]).max() < 2e-3 * 2000  # test_md_conservation.py bound
    assert 50.0 < th[-1, 0] < 1000.0  # the 250-atom cell is a thermalised snapshot, not a perfect lattice
    assert os.path.exists(os.path.join(wd, "restart.xyz")) and os.path.exists(os.path.join(wd, "neighbor.out"))


@pytest.mark.gpu
@pytest.mark.parametrize("ens", ["nvt_ber 300 600 20", "nvt_nhc 300 600 50", "nvt_bdp 300 600 50"])
def test_nvt_run_heats_towards_the_target(tmp_path, ens):
    """`ensemble nvt_ber|nvt_nhc T1 T2 Tc` through gpumd-mi: the temperature follows the ramp."""
    wd = _workdir(tmp_path, "replicate 2 2 2\npotential NEP\nvelocity 300 seed 42\nensemble %s\ntime_step 1\n"
                            "dump_thermo 50\nrun 1000\n" % ens)
    out = subprocess.run([EXE], cwd=wd, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    th = np.loadtxt(os.path.join(wd, "thermo.out"))
    assert th.shape == (20, 18)
    # model.xyz is a thermalised snapshot (T settles near 500 K within the first records whatever the
    # initial velocities); the thermostat holds the last quarter near the ramp's 525-600 K
    assert 450.0 < th[-5:, 0].mean() < 680.0


@pytest.mark.gpu
def test_dump_xyz_full_option_set(tmp_path):
    """dump_xyz with every quantity, a group selection and one file per frame (dump_xyz.cu:300-420), plus the
    group columns of restart.xyz (dump_restart.cu:111-131)."""
    wd, src = _grouped_workdir(
        tmp_path, "replicate 2 2 2\npotential NEP\nvelocity 600 seed 7\nensemble nve\ntime_step 2\n"
                  "dump_xyz 20 all.xyz mass charge velocity force potential unwrapped_position virial group_labels "
                  "precision double\n"
                  "dump_xyz 20 f_* group 1 2 unwrapped_position group_labels precision single\n"
                  "dump_restart 40\nrun 40\n", shift=-0.4)
    out = subprocess.run([EXE], cwd=wd, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    frames = H.read_xyz_frames(os.path.join(wd, "all.xyz"))
    assert len(frames) == 2 and frames[0]["n"] == 2000
    a = frames[1]
    assert a["comment"]["properties"] == ("species:S:1:pos:R:3:mass:R:1:charge:R:1:vel:R:3:forces:R:3:energy_atom:R:1:"
                                          "unwrapped_position:R:3:virial:R:9:group:I:2")
    np.testing.assert_array_equal(a["charge"][:, 0], np.where(np.array(a["species"]) == "Pb", 0.5, -0.5))
    np.testing.assert_array_equal(a["group"][:, 0], np.where(np.array(a["species"]) == "Te", 0, 1))
    np.testing.assert_array_equal(a["group"][:250, 1], np.where(np.arange(250) >= 240, 2, np.arange(250) % 2))
    # unwrapped - wrapped is a lattice vector; the cell was shifted, so atoms sit outside it from the start
    lat = a["lattice"]  # rows a, b, c
    frac = np.linalg.solve(lat.T, (a["unwrapped_position"] - a["pos"]).T)
    assert np.abs(frac - np.rint(frac)).max() < 1e-9 and np.abs(np.rint(frac)).max() >= 1
    # ... and stays the continuous trajectory: close to the (shifted, replicated) input coordinates
    _, _, pos0 = H.replicate(src["h"], np.zeros(250, np.int32), src["pos"] - 0.4, (2, 2, 2))
    assert np.abs(a["unwrapped_position"] - pos0).max() < 1.5
    # one file per frame, atoms of group 2 of grouping method 1 only, ascending atom index
    assert not os.path.exists(os.path.join(wd, "f_"))
    g = H.read_xyz_frames(os.path.join(wd, "f_40"))
    sel = np.nonzero(a["group"][:, 1] == 2)[0]
    assert len(g) == 1 and g[0]["n"] == len(sel) == 80
    assert g[0]["comment"]["properties"] == "species:S:1:pos:R:3:unwrapped_position:R:3:group:I:2"
    np.testing.assert_allclose(g[0]["pos"], a["pos"][sel], rtol=2e-8)
    np.testing.assert_allclose(g[0]["unwrapped_position"], a["unwrapped_position"][sel], rtol=2e-8)
    assert g[0]["comment"]["energy"] == "%.9g" % float(a["comment"]["energy"])
    r = H.read_xyz_frames(os.path.join(wd, "restart.xyz"))[0]
    assert r["comment"]["properties"] == "species:S:1:pos:R:3:mass:R:1:vel:R:3:group:I:2"
    np.testing.assert_array_equal(r["group"], a["group"])
    np.testing.assert_array_equal(r["pos"], a["pos"])


@pytest.mark.gpu
def test_reference_carbon_trajectory_golden(tmp_path):
    """The reference's one trajectory-level golden (tests/gpumd/carbon: 64,000-atom amorphous carbon, C_2022_NEP4, 100 NVE
    steps of 1 fs, `velocity 300` on glibc's default rand() stream, thermo1.out written by a DEBUG build of the CUDA
    code): gpumd-mi on the same run.in / model.xyz reproduces all ten thermo rows -- velocity initialisation,
    velocity-Verlet, the NEP force path and find_thermo pinned at the MD level against the reference itself."""
    g = np.load(H.golden("C", "carbon_64000.npz"))
    pos = g["pos"]
    with open(tmp_path / "model.xyz", "w") as f:
        f.write("%d\n%s\n" % (len(pos), str(g["header"])))
        for p in pos:
            f.write("C %.17g %.17g %.17g\n" % (p[0], p[1], p[2]))
    run_in = open(H.golden("C", "carbon_run.in")).read()
    assert "potentials/nep/C_2022_NEP4.txt" in run_in and "velocity        300" in run_in
    (tmp_path / "run.in").write_text(run_in.replace("../../../potentials/nep/C_2022_NEP4.txt", H.golden("C", "nep.txt")))
    out = subprocess.run([EXE], cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    got, ref = np.loadtxt(tmp_path / "thermo.out"), np.loadtxt(H.golden("C", "carbon_thermo1.out"))
    assert got.shape == ref.shape == (10, 18)
    # measured on MI355X: T and KE agree to 2e-7, PE to 3e-7 (0.05 eV of 1.6e5: FP32 summation order), stresses to 1e-4 GPa
    np.testing.assert_allclose(got[:, 0], ref[:, 0], rtol=2e-6)            # T
    np.testing.assert_allclose(got[:, 1], ref[:, 1], rtol=2e-6)            # KE
    np.testing.assert_allclose(got[:, 2], ref[:, 2], rtol=2e-6)            # PE
    np.testing.assert_allclose(got[:, 3:9], ref[:, 3:9], rtol=1e-3, atol=5e-4)  # stresses (GPa)
    np.testing.assert_allclose(got[:, 9:], ref[:, 9:], rtol=1e-12)         # box


def test_velocity_stream_is_glibc_rand(tmp_path):
    """Velocity::initialize draws from rand() (velocity.cu:40-75); the host restates glibc's generator so that nothing
    else in the process (HIP runtime, RCCL) can disturb the stream: it must BE glibc's, number for number."""
    import ctypes
    exe = os.path.join(H.ROOT, "tests", "emu", "gpumd-mi-emu")
    subprocess.run(["make", "-s", "-C", os.path.join(H.ROOT, "tests", "emu"), "all"], check=True)
    libc = ctypes.CDLL("libc.so.6")
    for seed in (1, 42, 123456789, 2 ** 31 + 5):
        libc.srand(ctypes.c_uint(seed))
        ref = [libc.rand() for _ in range(2000)]
        out = subprocess.run([exe, "--rand-check", "2000", str(seed)], capture_output=True, text=True)
        assert [int(x) for x in out.stdout.split()] == ref, seed
