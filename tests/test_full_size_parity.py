"""Parity where the box is largest: all three neighbour lists entry for entry and the forces,
energies and virials of every atom against the oracle, at BASELINE config 3's full size (PbTe 1,024,000 atoms, the
triclinic `replicate 16 16 16` cell and the orthogonal rock-salt variant) and at >= 250,000 atoms of config 4's and
config 5's models -- GPU tier; the CPU tier runs bars with the same box length (the fixed-point band is as wide)."""
import pytest

import helpers as H
import parity_cases as P


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["PbTe-1M-triclinic", "PbTe-1M-orthogonal", "UNEP-256k", "C-262k", "PbTe-bar-64k"])
def test_full_size_parity_gpu(name):
    P.check_full_size_parity(H.GpuDriver(), name)


@pytest.mark.parametrize("name", ["PbTe-bar-64k", "PbTe-ortho-bar"])
def test_long_box_parity_emulator(name):
    P.check_full_size_parity(H.EmuDriver(), name)


@pytest.mark.gpu
def test_carbon_8m_spot_parity_gpu():
    """Config 5's per-GPU share (8,000,000 carbon atoms, C_2022_NEP4): where N * ld index arithmetic and
    the buffers are largest.  A rattled periodic 8,000-atom diamond block tiled 10 x 10 x 10: every image of an atom sees the
    same environment, so (a) the list counts of ALL 8 M atoms must repeat the oracle's counts of the block, (b) the forces and
    energies of 27 interior blocks (216,000 atoms) must equal the oracle's on the periodic block within the stated tolerances,
    (c) the blocks agree with each other, (d) the total force vanishes; then the same through the scatter form of the force
    assembly (the form the run loops take)."""
    import numpy as np
    drv = H.GpuDriver()
    nep = H.golden("C", "nep.txt")
    h1, typ1, x1 = H.diamond((10, 10, 10), 3.57, rattle=0.03, seed=808)
    n1 = len(typ1)
    assert n1 == 8000
    orc = H.Oracle(nep)
    pe64, f64, _ = orc.compute(typ1, h1, x1, precision=64, path=0)
    L = orc.lists(typ1, h1, x1, path=0)
    h, typ, pos = H.replicate(h1, typ1, x1.reshape(3, n1).T, (10, 10, 10))
    n = len(typ)
    assert n == 8000000
    x = H.soa(pos)
    del pos
    eng = drv.engine(drv.model(nep), n)
    eng.set_win_lanes(1)
    for form in (0, 1):
        eng.set_force_form(form)
        _, pe, f, v = H.engine_force(drv, eng, h, typ.astype(np.int32), x)
        assert eng.stats().radial_tiles >= 1
        if form == 1:
            assert "lds_scatter_of_own_halves" in eng.describe(), eng.describe()
        F = f.reshape(3, 1000, n1)
        inner = [i * 100 + j * 10 + k for i in range(4, 7) for j in range(4, 7) for k in range(4, 7)]
        ref = f64.reshape(3, 1, n1)
        d = np.abs(F[:, inner, :] - ref) - 1e-4 * np.abs(ref)
        assert d.max() <= 3e-5, "form %d: interior blocks vs FP64 oracle, worst excess %.3e" % (form, d.max())
        np.testing.assert_allclose(pe.reshape(1000, n1)[inner], np.broadcast_to(pe64, (27, n1)), rtol=1e-5, atol=2e-5)
        # every block against an interior one: blocks at the periodic faces carry the FP32 rounding of a ~700 A minimum image
        assert np.abs(F - F[:, inner[13]:inner[13] + 1, :]).max() < 1e-3
        assert np.abs(f.reshape(3, n).sum(axis=1)).max() < (1e-6 if form == 1 else 1e-4 * np.sqrt(n))
        del F, f, pe, v
    for which, key in ((2, "skin"), (0, "radial"), (1, "angular")):
        onn = L[key][0]
        mx, nn, _ = H.engine_lists(drv, eng, n, which, ld=1)
        assert mx == onn.max()
        assert np.array_equal(nn.reshape(1000, n1), np.broadcast_to(onn, (1000, n1))), key
