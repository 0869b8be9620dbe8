"""CPU tier: pin the oracle (oracle/nep_oracle.c) against every known answer the reference holds
for this path and against the reference's own NEP_CPU compiled in place."""
import os

import numpy as np
import pytest

import helpers as H


def _frame0():
    fr = H.read_xyz_frames(H.golden("PbTe", "model.xyz"))[0]
    return fr


def test_pbte250_cuda_known_answer():
    """examples/gpumd_static/dump.xyz: energy, virial, forces written by the reference CUDA path
    (small-box branch, FP32).  check_force.m:9 expects ~1e-5 eV/A agreement."""
    orc = H.Oracle(H.golden("PbTe", "nep.txt"))
    fr = _frame0()
    typ = H.types_from_species(fr["species"], orc.symbols)
    x = H.soa(fr["pos"])
    ref = H.read_xyz_frames(H.golden("PbTe", "dump.xyz"))[0]
    for prec, ftol in ((32, 2e-5), (64, 2e-5)):
        pe, f, v = orc.compute(typ, fr["h"], x, precision=prec, path=-1)
        assert abs(pe.sum() - ref["energy"]) < 1e-5 * abs(ref["energy"])
        fo = f.reshape(3, -1).T
        assert np.abs(fo - ref["forces"]).max() < ftol
        # dump.xyz virial: 9 numbers xx xy xz yx yy yz zx zy zz (total)
        vt = v.reshape(9, -1).sum(axis=1)  # xx yy zz xy xz yz yx zx zy
        got = np.array([vt[0], vt[3], vt[4], vt[6], vt[1], vt[5], vt[7], vt[8], vt[2]])
        np.testing.assert_allclose(got, ref["virial"], rtol=1e-4, atol=2e-3)


def test_nep_prediction_frames():
    """examples/nep_prediction/*_train.out (written by the `nep` executable), all 25 frames."""
    orc = H.Oracle(H.golden("PbTe", "nep.txt"))
    frames = H.read_xyz_frames(H.golden("PbTe", "train_25frames.xyz"))
    out = np.load(H.golden("PbTe", "train_25frames_out.npz"))
    assert len(frames) == 25 and out["energy"].shape[0] == 25
    for k, fr in enumerate(frames):
        typ = H.types_from_species(fr["species"], orc.symbols)
        pe, f, v = orc.compute(typ, fr["h"], H.soa(fr["pos"]), precision=64)
        n = fr["n"]
        assert abs(pe.sum() / n - out["energy"][k, 0]) < 2e-5
        np.testing.assert_allclose(f.reshape(3, -1).T, out["force"][k * n:(k + 1) * n, :3], atol=2e-5)
        vt = v.reshape(9, -1).sum(axis=1) / n
        np.testing.assert_allclose(vt[[0, 1, 2, 3, 5, 4]], out["virial"][k, :6], rtol=2e-4, atol=2e-5)


def test_bazro3_golden_regression():
    """tests_pytest/fixtures/golden/bulk_bazro3.npz (energy, forces, ASE-sign stress)."""
    orc = H.Oracle(H.golden("BaZrO3", "nep.txt"))
    fr = H.read_xyz_frames(H.golden("BaZrO3", "BaZrO3-nat40-rattled.xyz"))[0]
    typ = H.types_from_species(fr["species"], orc.symbols)
    g = np.load(H.golden("BaZrO3", "bulk_bazro3.npz"))
    pe, f, v = orc.compute(typ, fr["h"], H.soa(fr["pos"]), precision=64)
    assert abs(pe.sum() - float(g["energy"])) < 1e-5 * abs(float(g["energy"]))
    np.testing.assert_allclose(f.reshape(3, -1).T, g["forces"], rtol=1e-4, atol=2e-5)
    vol = abs(np.linalg.det(fr["lattice"]))
    vt = v.reshape(9, -1).sum(axis=1)
    stress = -np.array([vt[0], vt[1], vt[2], vt[5], vt[4], vt[3]]) / vol  # ASE voigt xx yy zz yz xz xy
    np.testing.assert_allclose(stress, g["stress"], rtol=1e-4, atol=1e-6)


@pytest.mark.skipif(not H.ref_available(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("model,ntypes", [("PbTe/nep.txt", None), ("PbTe/nep_B.txt", None), ("C/nep.txt", 1),
                                           ("C/nep3.txt", 1), ("UNEP/nep.txt", 16), ("BaZrO3/nep.txt", 3),
                                           ("water/nep.txt", 2), ("Si/nep_3body.txt", 1), ("Si/nep_4body.txt", 1),
                                           ("Si/nep_5body.txt", 1), ("C/nep_2024.txt", 1)])
def test_oracle_vs_reference_nep_cpu(model, ntypes):
    """f64 oracle == the reference's vendored NEP_CPU (compiled in place) to ~1e-12."""
    nep = H.golden(*model.split("/"))
    orc = H.Oracle(nep)
    h, typ, x = H.pbte_supercell((2, 2, 2), num_types=ntypes, symbols=orc.symbols)
    ref = H.RefNepCpu(nep)
    pe_r, f_r, v_r = ref.compute(typ, h, x)
    pe, f, v = orc.compute(typ, h, x, precision=64, path=0)
    np.testing.assert_allclose(pe, pe_r, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(f, f_r, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(v, v_r, rtol=1e-9, atol=1e-9)
    # neighbour sets of the reference CPU code == oracle lists
    L = orc.lists(typ, h, x, path=0)
    for which, name in ((0, "radial"), (1, "angular")):
        nn_r, nl_r = ref.neighbors(len(typ), which)
        assert H.neighbor_sets(nn_r, nl_r) == H.neighbor_sets(*L[name])


def test_small_box_path_choice():
    """NEP::compute picks the small-box branch iff a periodic thickness <= 2.5 (rc + 1)."""
    orc = H.Oracle(H.golden("PbTe", "nep.txt"))
    fr = _frame0()
    typ = H.types_from_species(fr["species"], orc.symbols)
    assert orc.lists(typ, fr["h"], H.soa(fr["pos"]))["path"] == 1
    h, typ2, x = H.pbte_supercell((2, 2, 2))
    assert orc.lists(typ2, h, x)["path"] == 0


def _oracle_rows(L, flags, s, Fp, r12, fn, fnp):
    """The oracle's find_q / accumulate_f12 (FP64) for every radial order of one case."""
    import ctypes as C
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    nA1 = s.shape[0]
    has = np.ascontiguousarray(flags, np.int32)
    s64, Fp64, r64 = (np.ascontiguousarray(a, np.float64) for a in (s, Fp, r12))
    d12 = float(np.float32(np.sqrt((r12.astype(np.float64) ** 2).sum())))
    q = np.zeros((10, nA1))
    f = np.zeros((nA1, 3))
    for n in range(nA1):
        L.nepo_rows_find_q(4, has.ctypes.data_as(ip), nA1, n, s64[n].ctypes.data_as(dp), q.ctypes.data_as(dp))
        L.nepo_rows_accumulate_f12(4, has.ctypes.data_as(ip), n, nA1, C.c_double(d12), r64.ctypes.data_as(dp),
                                   C.c_double(float(fn)), C.c_double(float(fnp)), Fp64.ctypes.data_as(dp),
                                   s64[n].ctypes.data_as(dp), f[n].ctypes.data_as(dp))
    return q, f


def test_angular_rows_against_the_reference_functions():
    """Every invariant row -- 3-body L = 1..4, 222, 1111 and the extra 4-body rows 112 / 123 / 233 / 134 -- and
    its partial force against known answers of the reference's own find_q / accumulate_f12
    (nep_utilities.cuh:1523-1672, 1819-1947, compiled for the host; tests/golden/make_golden.py angular_rows).
    The reference computes in FP32: |dq| <= 2e-6 (1 + |q|), |df| <= 3e-5 (1 + |f|) on O(1..30) values."""
    g = np.load(H.golden("rows", "angular_rows_ref.npz"))
    L = H.oracle_lib()
    worst_q = worst_f = 0.0
    for c in range(g["s"].shape[0]):
        for k, fl in enumerate(g["flags"]):
            q, f = _oracle_rows(L, fl, g["s"][c], g["Fp"][c], g["r12"][c], g["fn"][c], g["fnp"][c])
            rows = 4 + int(fl.sum())
            qr, fr = g["q"][c, k].astype(np.float64), g["f12"][c, k].astype(np.float64)
            assert np.all(qr[rows:] == 0) and np.all(q[rows:] == 0)
            worst_q = max(worst_q, (np.abs(q - qr) / (1 + np.abs(qr))).max())
            worst_f = max(worst_f, (np.abs(f - fr) / (1 + np.abs(fr))).max())
    assert worst_q < 2e-6 and worst_f < 3e-5, (worst_q, worst_f)


@pytest.mark.skipif(not os.path.exists(os.path.join(H.ORACLE_DIR, "_ref", "libnep_utils_ref.so")),
                    reason="oracle/_ref not built (no /root/reference here)")
def test_angular_rows_fixture_is_what_the_reference_functions_return():
    """The committed vectors are reproduced by the live reference build (guards the fixture itself)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", H.golden("make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        mg.angular_rows(os.path.join(tmp, "rows.npz"))
        new = np.load(os.path.join(tmp, "rows.npz"))
        ref = np.load(H.golden("rows", "angular_rows_ref.npz"))
        for k in ref.files:
            np.testing.assert_array_equal(new[k], ref[k])


_THERMO_REF = os.path.join(H.ROOT, "oracle", "_ref", "libthermo_ref.so")


@pytest.mark.skipif(not os.path.exists(_THERMO_REF), reason="oracle/_ref not built (no /root/reference here)")
def test_nose_hoover_chain_vs_reference_nhc():
    """oracle nepo_nhc == the reference's own nhc() (src/integrate/ensemble_nhc.cu:102-164, compiled in place by
    oracle/ref_thermo_wrap.cpp): scale factors and the chain state over a driven sequence of kinetic energies."""
    import ctypes as C
    R = C.CDLL(_THERMO_REF)
    R.nepref_nhc.restype = C.c_double
    R.nepref_nhc.argtypes = [C.c_int, H._dp, H._dp, H._dp, C.c_double, C.c_double, C.c_double, C.c_double]
    L = H.oracle_lib()
    n, t_coup, dt = 2000, 50.0, 1.0 / H.TIME_UNIT
    st = np.zeros(12)
    L.nepo_nhc_init(n, 300.0, t_coup, dt, H._p(st, H._dp))
    pos, vel, mas = st[0:4].copy(), st[4:8].copy(), st[8:12].copy()
    rng = np.random.default_rng(5)
    dN = 3.0 * n
    for step in range(200):
        T_now = 300.0 + 150.0 * np.sin(0.07 * step) + rng.normal(0, 5)
        target = 300.0 + 0.5 * step
        ek2, kT = T_now * dN * H.K_B, H.K_B * target
        f_o = L.nepo_nhc(H._p(st, H._dp), ek2, kT, dN, 0.5 * dt)
        f_r = R.nepref_nhc(4, H._p(pos, H._dp), H._p(vel, H._dp), H._p(mas, H._dp), ek2, kT, dN, 0.5 * dt)
        assert f_o == f_r, (step, f_o, f_r)  # the same double arithmetic in the same order: bit for bit
    assert np.array_equal(st[0:4], pos) and np.array_equal(st[4:8], vel) and np.array_equal(st[8:12], mas)
    assert np.abs(vel).max() > 0 and abs(f_o - 1.0) > 1e-9  # the chain really acted


@pytest.mark.skipif(not os.path.exists(_THERMO_REF), reason="oracle/_ref not built (no /root/reference here)")
def test_bdp_draws_vs_reference_resamplekin():
    """oracle nepo_bdp_factor == the reference's own resamplekin / gasdev / gamdev (src/integrate/svr_utilities.cuh)
    driven by std::mt19937 with the same seed, draw for draw.  gasdev keeps a function-local cache in the reference,
    so the comparison runs in a process of its own."""
    code = r'''
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import helpers as H
R = C.CDLL(%r)
L = H.oracle_lib()
n, seed, tc = 250, 20240924, 25.0
T = 300.0 + 80.0 * np.sin(0.3 * np.arange(64))
out = np.zeros(64)
R.nepref_bdp_factors(C.c_uint(seed), 64, n, H._p(T, H._dp), C.c_double(350.0), C.c_double(tc), H._p(out, H._dp))
rng = C.create_string_buffer(L.nepo_bdp_sizeof())
L.nepo_bdp_seed(rng, seed)
L.nepo_bdp_factor.restype = C.c_double
got = np.array([L.nepo_bdp_factor(rng, n, C.c_double(t), C.c_double(350.0), C.c_double(tc)) for t in T])
assert np.array_equal(got, out), np.abs(got - out).max()
assert np.abs(out - 1.0).max() > 1e-3
# small systems take the other branches of gamdev / resamplekin_sumnoises (ia < 6, odd / even degrees of freedom)
print("ok")
''' % (H.ROOT, os.path.join(H.ROOT, "tests"), _THERMO_REF)
    import subprocess
    import sys
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout + res.stderr
    for n_atoms in (1, 2, 3, 4, 5):  # 3, 6, ... 15 degrees of freedom: the ia < 6 product branch and both parities
        code2 = code.replace("n, seed, tc = 250, 20240924, 25.0", "n, seed, tc = %d, 77, 3.0" % n_atoms)
        res = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True)
        assert res.returncode == 0 and "ok" in res.stdout, (n_atoms, res.stdout + res.stderr)
