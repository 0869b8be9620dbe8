"""dump_xyz through gpumd-mi, case for case what the reference's own suite asks of it
(tests_pytest/test_io_dump_commands.py:116-135,145-400,442-510), on the PbTe 250-atom cell."""
import os
import re
import subprocess

import numpy as np
import pytest

import helpers as H

EXE = os.path.join(H.ROOT, "gpumd_amd", "bin", "gpumd-mi")
N_STEPS = 5  # io_helpers.py BASE_N_STEPS


@pytest.fixture(scope="module", autouse=True)
def _build():
    if not os.path.exists(EXE):
        import __graft_entry__ as g
        g.build()


def _run(tmp_path, lines, n_groups=0, check_only=False):
    """run_command_io_case (io_helpers.py:70-130): potential, velocity, ensemble nve, time_step 1, dump_thermo 1,
    the case's lines, run 5; n_groups = 1: everyone in group 0, 2: first half 0 / second half 1."""
    src = open(H.golden("PbTe", "model.xyz")).read().split("\n")
    n = int(src[0])
    if n_groups:
        src[1] = src[1].rstrip() + ":group:I:1"
        for k in range(n):
            src[2 + k] = src[2 + k].rstrip() + " %d" % (0 if n_groups == 1 or k < n // 2 else 1)
    (tmp_path / "model.xyz").write_text("\n".join(src))
    run_in = ["potential " + H.golden("PbTe", "nep.txt"), "velocity 300 seed 3", "ensemble nve", "time_step 1",
              "dump_thermo 1"] + lines + ["run %d" % N_STEPS]
    (tmp_path / "run.in").write_text("\n".join(run_in) + "\n")
    return subprocess.run([EXE] + (["--check-input"] if check_only else []), cwd=str(tmp_path), capture_output=True,
                          text=True), n


def _comment_fields(path):
    fields = {}
    for m in re.finditer(r'(\w+)=(?:"([^"]*)"|(\S+))', open(path).read().split("\n")[1]):
        fields[m.group(1)] = (m.group(2) if m.group(2) is not None else m.group(3)).split()
    return fields


def _digits(token):
    mantissa = token.split("e")[0].split("E")[0]
    return len(mantissa.replace("-", "").replace(".", "").lstrip("0"))


def _row_digits(path, n):
    rows = open(path).read().split("\n")[2:2 + n]
    return max(_digits(t) for r in rows for t in r.split()[1:])


@pytest.mark.gpu
def test_comment_line_is_well_formed(tmp_path):
    res, n = _run(tmp_path, ["dump_xyz 1 comment.xyz virial"])
    assert res.returncode == 0, res.stdout
    path = tmp_path / "comment.xyz"
    f = _comment_fields(path)
    line = open(path).read().split("\n")[1]
    for key in ("Lattice", "virial", "stress"):
        assert len(f[key]) == 9
        assert not re.search(key + '="([^"]*)"', line).group(1).startswith(" ")
    frames = H.read_xyz_frames(str(path))
    assert len(frames) == N_STEPS and frames[0]["n"] == n
    np.testing.assert_allclose(frames[0]["lattice"], H.read_xyz_frames(H.golden("PbTe", "model.xyz"))[0]["lattice"],
                               atol=1e-6)
    thermo = np.loadtxt(tmp_path / "thermo.out")
    np.testing.assert_allclose(float(f["energy"][0]), thermo[0, 2], rtol=1e-6)  # U of the same step
    for key in ("virial", "stress"):
        t = np.array(f[key], dtype=float).reshape(3, 3)
        np.testing.assert_array_equal(t, t.T)
    # stress on the comment line = thermo.out's pressure columns (GPa) / 160.2177
    s = np.array(f["stress"], dtype=float).reshape(3, 3)
    np.testing.assert_allclose([s[0, 0], s[1, 1], s[2, 2], s[1, 2], s[0, 2], s[0, 1]], thermo[0, 3:9] / 160.2177,
                               rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
def test_precision_reaches_the_comment_line_and_default_is_single(tmp_path):
    res, n = _run(tmp_path, ["dump_xyz 1 default.xyz virial force", "dump_xyz 1 explicit.xyz virial force precision single",
                             "dump_xyz 1 double.xyz virial force precision double"])
    assert res.returncode == 0, res.stdout
    assert (tmp_path / "default.xyz").read_text() == (tmp_path / "explicit.xyz").read_text()
    single, double = _comment_fields(tmp_path / "default.xyz"), _comment_fields(tmp_path / "double.xyz")
    for key in ("energy", "virial", "stress"):
        sd = max(_digits(t) for t in single[key])
        assert sd <= 9 and max(_digits(t) for t in double[key]) > sd
    assert _row_digits(tmp_path / "default.xyz", n) <= 9 < _row_digits(tmp_path / "double.xyz", n)
    # nine significant digits whatever the magnitude (the reason %g replaced a fixed %.8f)
    a = H.read_xyz_frames(str(tmp_path / "default.xyz"))[0]["forces"]
    b = H.read_xyz_frames(str(tmp_path / "double.xyz"))[0]["forces"]
    nz = np.abs(b) > 0
    assert (np.abs(a - b)[nz] / np.abs(b)[nz]).max() < 1e-8


@pytest.mark.gpu
def test_per_atom_virial_adds_up_to_the_total_and_group_selects_atoms(tmp_path):
    res, n = _run(tmp_path, ["dump_xyz 1 virial.xyz virial precision double group 0 1"], n_groups=2)
    assert res.returncode == 0, res.stdout
    path = str(tmp_path / "virial.xyz")
    fr = H.read_xyz_frames(path)[0]
    assert fr["n"] == n - n // 2  # only group 1 is written
    rows = np.array([[float(x) for x in r.split()[4:13]] for r in open(path).read().split("\n")[2:2 + fr["n"]]])
    # the header keeps describing the whole system: the second half alone must NOT add up to it ...
    total = np.array(_comment_fields(path)["virial"], dtype=float).reshape(3, 3)
    assert np.abs(rows.reshape(-1, 3, 3).sum(axis=0) - total).max() > 1e-3
    # ... and with everyone in one group it does (column order xx xy xz yx yy yz zx zy zz, six independent sums)
    sub = tmp_path / "all"
    sub.mkdir()
    res, n = _run(sub, ["dump_xyz 1 virial.xyz virial precision double group 0 0"], n_groups=1)
    assert res.returncode == 0, res.stdout
    path = str(sub / "virial.xyz")
    rows = np.array([[float(x) for x in r.split()[4:13]] for r in open(path).read().split("\n")[2:2 + n]])
    per_atom = rows.reshape(-1, 3, 3).sum(axis=0)
    total = np.array(_comment_fields(path)["virial"], dtype=float).reshape(3, 3)
    for i, j in ((0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)):
        np.testing.assert_allclose(per_atom[i, j], total[i, j], rtol=1e-6, atol=1e-8)


@pytest.mark.gpu
def test_column_order_does_not_follow_argument_order(tmp_path):
    q = ["mass", "velocity", "force", "potential", "virial"]
    res, _ = _run(tmp_path, ["dump_xyz 1 forward.xyz " + " ".join(q), "dump_xyz 1 reversed.xyz " + " ".join(q[::-1])])
    assert res.returncode == 0, res.stdout
    assert (tmp_path / "forward.xyz").read_text() == (tmp_path / "reversed.xyz").read_text()
    assert _comment_fields(tmp_path / "forward.xyz")["Properties"][0] == (
        "species:S:1:pos:R:3:mass:R:1:vel:R:3:forces:R:3:energy_atom:R:1:virial:R:9")


@pytest.mark.gpu
def test_group_labels_match_the_groupings(tmp_path):
    res, n = _run(tmp_path, ["dump_xyz 1 labels.xyz group_labels"], n_groups=2)
    assert res.returncode == 0, res.stdout
    assert _comment_fields(tmp_path / "labels.xyz")["Properties"][0].endswith(":group:I:1")
    labels = H.read_xyz_frames(str(tmp_path / "labels.xyz"))[0]["group"].reshape(-1)
    np.testing.assert_array_equal(labels, [0] * (n // 2) + [1] * (n - n // 2))


@pytest.mark.gpu
def test_one_file_per_frame_for_a_starred_name(tmp_path):
    res, n = _run(tmp_path, ["dump_xyz 1 frame.xyz*"])
    assert res.returncode == 0, res.stdout
    produced = sorted(p.name for p in tmp_path.glob("frame.xyz*"))
    assert produced == ["frame.xyz%d" % s for s in range(1, N_STEPS + 1)]
    for name in produced:
        frames = H.read_xyz_frames(str(tmp_path / name))
        assert len(frames) == 1 and frames[0]["n"] == n


@pytest.mark.gpu
def test_restart_file_holds_one_full_frame(tmp_path):
    res, n = _run(tmp_path, ["dump_restart 1"])
    assert res.returncode == 0, res.stdout
    frames = H.read_xyz_frames(str(tmp_path / "restart.xyz"))
    assert len(frames) == 1 and frames[0]["n"] == n and frames[0]["vel"].shape == (n, 3)


# test_io_dump_commands.py:442-456 (the substring matters as much as the exit code)
@pytest.mark.parametrize("args,msg", [
    ("1 f.xyz forcee", "Unrecognized argument"),
    ("1 f.xyz group 0 0 group 0 0", "more than once"),
    ("1 f.xyz precision double precision single", "more than once"),
    ("1 f.xyz force force", "more than once"),
    ("1 f.xyz precision triple", "Invalid precision"),
    ("1 f.xyz velocity group", "group_labels"),
    ("1 f.xyz group 0", "group_labels"),
    ("1 f.xyz group -1 0", "Grouping method"),
    ("1", "at least 2 parameters"),
    ("0 f.xyz", "dump interval")])
def test_invalid_dump_xyz_arguments_are_rejected(tmp_path, args, msg):
    res, _ = _run(tmp_path, ["dump_xyz " + args], n_groups=1, check_only=True)
    assert res.returncode != 0 and msg in res.stdout + res.stderr, res.stdout


# test_io_dump_commands.py:488-510
@pytest.mark.parametrize("line", ["dump_position 1", "dump_velocity 1", "dump_force 1", "dump_exyz 1 1 1 1",
                                  "dump_xyz -1 0 1 dump.xyz"])
def test_removed_commands_are_rejected(tmp_path, line):
    res, _ = _run(tmp_path, [line], check_only=True)
    assert res.returncode != 0 and "dump_xyz" in res.stdout + res.stderr, res.stdout
