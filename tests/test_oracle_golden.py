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
    """examples/nep_prediction/*_train.out (written by the `nep` executable), first 2 frames."""
    orc = H.Oracle(H.golden("PbTe", "nep.txt"))
    frames = H.read_xyz_frames(H.golden("PbTe", "train_2frames.xyz"))
    out = np.load(H.golden("PbTe", "train_2frames_out.npz"))
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
