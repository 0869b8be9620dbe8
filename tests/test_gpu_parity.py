"""GPU tier (-m gpu): the product library libnepmi.so on an MI355X, through the C ABI, against the
oracle on the same seeded inputs, plus size-independent properties at BASELINE.json's full size."""
import numpy as np
import pytest

import helpers as H
import parity_cases as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def drv():
    return H.GpuDriver()


@pytest.mark.parametrize("name", list(P.MODELS))
def test_force_parity(drv, name):
    P.check_force_parity(drv, name)


@pytest.mark.parametrize("name", ["PbTe-A", "UNEP-v1", "C-2022"])
def test_force_parity_generic_shape(drv, name):
    P.check_force_parity(drv, name, generic=True, check_lists=False)


@pytest.mark.parametrize("name", ["PbTe-A", "PbTe-ortho", "UNEP-v1"])
def test_force_parity_without_lds_window(drv, name):
    """the plain gather radial kernel (fallback of the LDS-window pass) gives the same answers"""
    P.check_force_parity(drv, name, tiles=False)


@pytest.mark.parametrize("name,forced", [("Si-3body", False), ("Si-4body", False), ("water-model", True), ("PbTe-A", True),
                                         ("PbTe-ortho-big", True), ("C-2022", True)])
def test_force_parity_zero_padded_into_a_cover_shape(drv, name, forced, monkeypatch):
    """tests/test_emu_parity.py::test_force_parity_zero_padded_into_a_cover_shape on the real kernels: a model without compiled
    kernels of its own shape served, zero-padded, by a compiled cover shape -- against the oracle."""
    monkeypatch.setenv("NEPMI_JIT", "2")
    if forced:
        monkeypatch.setenv("NEPMI_FORCE_COVER", "1")
    eng = P.check_force_parity(drv, name, f32_atol=4e-5)
    assert "shape=cover(" in eng.describe() and "zero_padded_model" in eng.describe(), eng.describe()


@pytest.mark.parametrize("name", ["PbTe-A", "C-2022", "BaZrO3"])
def test_force_parity_without_mfma(drv, name):
    """The per-atom ANN kernel (taken automatically for many-type models such as UNEP-v1)."""
    P.check_force_parity(drv, name, check_lists=False, mfma=False)


@pytest.mark.parametrize("name", ["PbTe-A", "PbTe-B", "C-2022", "BaZrO3", "UNEP-v1-big"])
def test_mfma_ann_matches_per_atom_ann(drv, name):
    """Matrix-core ANN (mode 2) and the fused descriptor + ANN kernel (mode 1, where the shape allows it) vs the per-atom
    ANN kernel (mode 0): same contractions, different f32 summation order.  UNEP-v1 (16 types): the by-type form of the
    matrix-core kernel (one workgroup serves one type over sixteen chunks; no radial-table rows -- it runs where the force
    assembly contracts from the atoms' radial Fp rows, i.e. on the one-lane window kernels)."""
    nep_rel, build, _ = P.MODELS[name]
    nep = H.golden(*nep_rel.split("/"))
    h, typ, x = build()
    n = len(typ)
    model = drv.model(nep)
    out = []
    for on in (2, 0, 1):
        eng = drv.engine(model, n)
        eng.set_mfma(on)
        if name.startswith("UNEP"):
            eng.set_win_lanes(1)
        _, pe, f, v = H.engine_force(drv, eng, h, typ, x)
        if name.startswith("UNEP"):
            assert ("ann=mfma_f32_32x32x2(one_type_per_workgroup)" in eng.describe()) == (on != 0), eng.describe()
        q = drv.zeros(model.info.dim * n, dtype=np.float32)
        fp = drv.zeros(model.info.dim * n, dtype=np.float32)
        eng.descriptors(q, fp)
        out.append((pe, f, v, drv.host(fp).reshape(-1, n)))
    pe0, f0, v0, fp0 = out[1]
    for pe1, f1, v1, fp1 in (out[0], out[2]):
        np.testing.assert_allclose(fp1, fp0, rtol=1e-4, atol=2e-6 * np.abs(fp0).max())
        np.testing.assert_allclose(pe1, pe0, rtol=1e-5, atol=1e-5)  # (a per-atom energy is a sum of terms up to ~10 eV: one FP32 ulp there is 1e-6)
        assert np.abs(f1 - f0).max() <= 1e-5 * max(1.0, np.abs(f0).max())
        assert np.abs(v1 - v0).max() <= 2e-5 * max(1.0, np.abs(v0).max())


@pytest.mark.parametrize("name", ["PbTe-A", "PbTe-B", "BaZrO3", "PbTe-ortho", "PbTe-3x3x3", "C-2022", "C-nep3", "water-model"])
def test_fused_angular_kernel_matches_the_separate_kernels(drv, name):
    """Angular descriptor + ANN + partial angular forces in one lane-pair kernel (nep_fused.h, the default where the
    descriptor + ANN fusion applies) vs the separate kernels: descriptor and Fp through the parity hook, energies, forces and
    virials; the same contractions, the ANN's dot products summed in another order."""
    nep_rel, build, _ = P.MODELS[name]
    nep = H.golden(*nep_rel.split("/"))
    h, typ, x = build()
    n = len(typ)
    model = drv.model(nep)
    out = []
    for fused in (True, False):
        eng = drv.engine(model, n)
        eng.set_angular_fused(fused)
        _, pe, f, v = H.engine_force(drv, eng, h, typ, x)
        assert ("partial_forces_in_one_kernel" in eng.describe()) == fused
        q = drv.zeros(model.info.dim * n, dtype=np.float32)
        fp = drv.zeros(model.info.dim * n, dtype=np.float32)
        eng.descriptors(q, fp)
        out.append((pe, f, v, drv.host(q).reshape(-1, n), drv.host(fp).reshape(-1, n)))
    pe1, f1, v1, q1, fp1 = out[0]
    pe0, f0, v0, q0, fp0 = out[1]
    np.testing.assert_allclose(q1, q0, rtol=1e-6, atol=1e-7 * np.abs(q0).max())
    np.testing.assert_allclose(fp1, fp0, rtol=1e-4, atol=2e-6 * np.abs(fp0).max())
    np.testing.assert_allclose(pe1, pe0, rtol=1e-5, atol=1e-5)
    assert np.abs(f1 - f0).max() <= 1e-5 * max(1.0, np.abs(f0).max())
    assert np.abs(v1 - v0).max() <= 2e-5 * max(1.0, np.abs(v0).max())


@pytest.mark.parametrize("lanes", [1, 2, 4])
@pytest.mark.parametrize("name", ["PbTe-A", "PbTe-B", "C-2022", "UNEP-v1", "BaZrO3"])
def test_force_parity_lanes_per_atom(drv, name, lanes):
    """The LDS-window kernels with 1, 2 or 4 lanes per atom (the engine picks by the number of bricks; these systems
    are small, so the default is 4): same lists bit for bit, same forces within the FP32 tolerance."""
    P.check_force_parity(drv, name, lanes=lanes)


@pytest.mark.parametrize("static", [True, False])
@pytest.mark.parametrize("name", ["PbTe-A", "PbTe-B", "C-2022", "UNEP-v1", "UNEP-v1-big", "BaZrO3", "PbTe-3x3x3"])
def test_force_parity_window_layouts(drv, name, static):
    """One lane per atom on the static window layout (Verlet entries kept as LDS slots, four to a word; two-type models
    walk list B as two type-pure streams) and on the scanned layout: the same lists bit for bit."""
    eng = P.check_force_parity(drv, name, lanes=1, win_static=static)
    if name == "UNEP-v1-big":  # many types: the neighbour's half of a pair force comes from its radial Fp row (static layout)
        assert ("neighbour_half_from_fp_rows" in eng.describe()) == bool(static)


@pytest.mark.parametrize("name", ["PbTe-A", "PbTe-B", "PbTe-3x3x3", "PbTe-ortho-big", "C-2022", "C-nep3", "C-2024", "Si-5body",
                                  "UNEP-v1-big", "UNEP-v1", "BaZrO3"])
def test_force_parity_scatter_form(drv, name):
    """The force assembly as an LDS-local scatter of the own pair halves into fixed-point window accumulators + the fold
    (gpumd_amd/csrc/nep_scatter.h; the form the run loops take), forced for a per-call evaluation -- which then adds the
    virial-only pass of the gather form, so every check of the parity case applies unchanged: energies, every force against
    the FP64 and the FP32 oracle, per-atom virials in the reference's attribution, descriptors, the three lists."""
    eng = P.check_force_parity(drv, name, lanes=1, win_static=True, force_form=1)
    applies = name != "BaZrO3"  # (three types: neither the type-pure segments of <= 2 types nor the Fp rows of > 4: gather form)
    if eng.stats().radial_tiles == 3 and applies:  # (smaller boxes have no window kernels at all: the gather kernels ran)
        assert "lds_scatter_of_own_halves" in eng.describe(), eng.describe()
    if name == "UNEP-v1-big":  # many types: the own half contracted per pair from the coefficient table in LDS, four lanes per atom
        assert eng.stats().radial_tiles == 3


def test_scatter_form_properties(drv):
    """Scatter form vs gather form of one engine on the same positions: forces equal to fixed-point + f32 rounding, the TOTAL
    virial of the own-half form equals the gather form's, total force exactly zero in fixed point (+g and -g are the same
    integer), a second call bit-identical (integer sums do not depend on the order the lanes arrive in)."""
    nep = H.golden("PbTe", "nep.txt")
    h, typ, x = H.pbte_supercell((3, 3, 3), rattle=0.05, seed=11)
    n = len(typ)
    model = drv.model(nep)
    out = {}
    for form in (0, 1):
        eng = drv.engine(model, n)
        eng.set_win_lanes(1)
        eng.set_force_form(form)
        _, pe, f, v = H.engine_force(drv, eng, h, typ, x)
        _, pe2, f2, v2 = H.engine_force(drv, eng, h, typ, x)
        assert np.array_equal(f, f2) and np.array_equal(pe, pe2) and np.array_equal(v, v2)
        out[form] = (pe, f, v, eng.describe())
    assert "lds_scatter_of_own_halves" in out[1][3] and "lds_scatter" not in out[0][3]
    assert np.array_equal(out[0][0], out[1][0])  # energies: the same kernels
    assert np.abs(out[0][1] - out[1][1]).max() < 4e-6, np.abs(out[0][1] - out[1][1]).max()
    np.testing.assert_allclose(out[1][2], out[0][2], rtol=0, atol=1e-12)  # per-atom virials: the same virial-only gather pass
    # total force: every pair half enters twice with opposite sign as the same integer of 2^-22 eV/A
    fsum = out[1][1].reshape(3, n).sum(axis=1)
    assert np.abs(fsum).max() < 1e-9, fsum


def test_scatter_guard_band_hands_over_to_the_gather_form(drv, monkeypatch):
    """The fixed-point sums of the scatter form hold +-512 eV/A; a pair half beyond 64 eV/A (or a net force beyond 256) makes the
    engine leave the form.  No NEP model of the repository produces such forces on a sane structure, so the test lowers the band
    through nepmi_engine_set_scatter_guard: (a) a force evaluation that leaves the band is REPEATED in the
    gather form before anything is returned -- bit-identical to an engine that never used the scatter form; (b) inside a run loop
    the step freezes like a skin trip and is re-run in the gather form -- same trajectory as the gather-form engine (to the
    rounding of a different list generation: the replay rebuilds the lists)."""
    nep = H.golden("PbTe", "nep.txt")
    h, typ, x = H.pbte_supercell((6, 6, 6), rattle=0.05, seed=21)
    mass = np.where(typ == 0, H.MASS["Te"], H.MASS["Pb"]).astype(np.float64)
    n = len(typ)
    model = drv.model(nep)

    def make(form, guard):
        eng = drv.engine(model, n)
        if guard:
            eng.set_scatter_guard(float(guard))
        eng.set_win_lanes(1)
        eng.set_force_form(form)
        return eng

    # (a) one evaluation
    ref = make(0, None)
    _, pe0, f0, v0 = H.engine_force(drv, ref, h, typ, x)
    assert np.abs(f0).max() > 0.5  # the lowered band is below the forces of this structure
    sc = make(1, None)
    H.engine_force(drv, sc, h, typ, x)
    assert "lds_scatter" in sc.describe()  # with the real band the scatter form stays
    low = make(1, "0.25")
    _, pe1, f1, v1 = H.engine_force(drv, low, h, typ, x)
    assert "lds_scatter" not in low.describe(), low.describe()
    assert np.array_equal(f1, f0) and np.array_equal(pe1, pe0) and np.array_equal(v1, v0)
    _, pe2, f2, v2 = H.engine_force(drv, low, h, typ, x)  # ... and for the rest of the engine's life
    assert "lds_scatter" not in low.describe() and np.array_equal(f2, f0)

    # (b) a run loop whose first force evaluation leaves the band (entered with zero forces: no evaluation before the loop)
    vel = H.maxwell_velocities(mass, 600.0, seed=5)
    out = []
    for eng in (make(0, None), make(1, "0.25")):
        d_t, d_m, d_x, d_v = drv.dev(typ), drv.dev(mass), drv.dev(x), drv.dev(vel)
        d_pe, d_f, d_w = drv.zeros(n), drv.zeros(3 * n), drv.zeros(9 * n)
        th = eng.run_nve(h, d_t, d_m, 1.0 / H.TIME_UNIT, 12, d_x, d_v, d_pe, d_f, d_w, thermo_every=4)
        out.append((drv.host(d_x), drv.host(d_v), drv.host(d_f), np.asarray(th), eng.describe(), eng.stats().discarded_steps))
    g, t = out
    assert "lds_scatter" not in t[4], t[4]
    assert t[5] > g[5]  # steps enqueued behind the frozen one ran as no-ops and were replayed
    assert np.abs(t[0] - g[0]).max() < 2e-8 and np.abs(t[1] - g[1]).max() < 2e-8 and np.abs(t[2] - g[2]).max() < 2e-5
    np.testing.assert_allclose(t[3], g[3], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("model", ["PbTe", "C"])
def test_run_loop_forms_are_bit_identical(drv, model):
    """The run loop's scatter-form steps with the per-step radial list as inside bits over the packed Verlet words (opt-in:
    option "radial_mask"), as a compacted list and as wave-synchronous words (the default): the same pairs with the same per-pair arithmetic into integer sums -- identical
    positions, velocities, forces and energies after 40 steps with list rebuilds, bit for bit; per-atom virials (the
    virial-only gather pass at the exit, which needs the compacted list rebuilt on demand) identical as well; and both
    against the gather form within f32 rounding."""
    if model == "PbTe":
        nep, (h, typ, x) = H.golden("PbTe", "nep.txt"), H.pbte_supercell((6, 6, 6), rattle=0.03, seed=17)
        mass = np.where(typ == 0, H.MASS["Te"], H.MASS["Pb"]).astype(np.float64)
    else:
        nep, (h, typ, x) = H.golden("C", "nep.txt"), H.diamond((18, 18, 18), 3.57, rattle=0.03, seed=18)
        mass = np.full(len(typ), H.MASS["C"])
    n = len(typ)
    vel = H.maxwell_velocities(mass, 2500.0, seed=4)
    m = drv.model(nep)
    out = []
    # (form 1: these systems are below the size the run loops' rule asks for)
    for form, mask, sync in ((1, True, False), (1, False, False), (0, True, False), (1, False, True)):
        eng = drv.engine(m, n)
        eng.set_win_lanes(1)
        eng.set_force_form(form)
        eng.set_radial_mask(mask)
        eng.set_radial_sync(sync)
        d_t, d_m, d_x, d_v = drv.dev(typ), drv.dev(mass), drv.dev(x), drv.dev(vel)
        d_pe, d_f, d_w = drv.zeros(n), drv.zeros(3 * n), drv.zeros(9 * n)
        eng.force_compute(h, d_t, d_x, d_pe, d_f, d_w)
        th = eng.run_nve(h, d_t, d_m, 2.0 / H.TIME_UNIT, 40, d_x, d_v, d_pe, d_f, d_w, thermo_every=10)
        out.append((drv.host(d_x), drv.host(d_v), drv.host(d_f), drv.host(d_pe), drv.host(d_w), np.asarray(th), eng.describe(),
                    eng.stats().num_rebuild))
    a, b, g, w = out
    assert "inside_bits" in a[6] and "compacted" in b[6] and "lds_scatter" in b[6] and "lds_scatter" not in g[6], (a[6], b[6], g[6])
    # ... and as wave-synchronous words (the default of the run loops; nep_window.h: SyncFifo): the same pairs once more, in words
    # every lane of a wavefront stores at the same time (padded with the sentinel slot) -- bit for bit as well
    assert "wave_synchronous_words" in w[6] and "lds_scatter" in w[6], w[6]
    for i in range(5):
        assert np.array_equal(w[i], b[i]), (i, np.abs(w[i] - b[i]).max())
    assert np.array_equal(w[5][:, :2], b[5][:, :2]) and w[7] == b[7]
    np.testing.assert_allclose(w[5], b[5], rtol=1e-6, atol=1e-9)
    if model == "PbTe":
        assert a[7] >= 2  # list rebuilds inside the run (the stiff diamond lattice keeps its lists over these 40 steps)
    for i in range(5):
        assert np.array_equal(a[i], b[i]), (i, np.abs(a[i] - b[i]).max())
    # thermo records: temperature and energy identical; the stresses come from the own-half virials, which are f32 sums over
    # the pairs in the order they are walked (list order vs word order)
    assert np.array_equal(a[5][:, :2], b[5][:, :2])
    np.testing.assert_allclose(a[5], b[5], rtol=1e-6, atol=1e-9)
    assert np.abs(a[0] - g[0]).max() < 1e-6 and np.abs(a[2] - g[2]).max() < 2e-4
    np.testing.assert_allclose(a[5][:, :2], g[5][:, :2], rtol=1e-6)


def test_virial_mode_totals_lets_per_call_evaluations_take_the_scatter_form(drv):
    """nepmi_engine_set_virial_mode(e, 1): a host that only needs the TOTAL virial (find_thermo) gets the run loops' rule on
    the per-call entry points -- forces and energies equal to f32 rounding, the SUM of the virial planes equal, per-atom planes in
    the own-half attribution (not compared); mode 0 afterwards: the reference's attribution again, bit for bit."""
    nep = H.golden("PbTe", "nep.txt")
    h, typ, x = H.rocksalt_orthogonal((30, 30, 30), rattle=0.02, seed=9)  # 216,000 atoms, about a thousand bricks (>= 768: asserted below)
    n = len(typ)
    model = drv.model(nep)
    eng = drv.engine(model, n)
    _, pe0, f0, v0 = H.engine_force(drv, eng, h, typ, x)
    assert "lds_scatter" not in eng.describe()
    eng.set_virial_mode(1)
    _, pe1, f1, v1 = H.engine_force(drv, eng, h, typ, x)
    assert int(eng.describe().split("bricks=")[1].split()[0]) >= 768 and "lds_scatter" in eng.describe(), eng.describe()
    np.testing.assert_allclose(pe1, pe0, rtol=1e-6, atol=1e-6)
    assert np.abs(f1 - f0).max() < 1e-5
    np.testing.assert_allclose(v1.reshape(9, n).sum(axis=1), v0.reshape(9, n).sum(axis=1), rtol=1e-5, atol=1e-3)
    eng.set_virial_mode(0)
    _, pe2, f2, v2 = H.engine_force(drv, eng, h, typ, x)
    assert np.array_equal(pe2, pe0) and np.array_equal(f2, f0) and np.array_equal(v2, v0)


def test_one_force_kernel_per_brick_matches_the_separate_kernels(drv):
    """nep_brick.h (opt-in: measured slower than the two kernels it replaces): fused angular kernel + scatter-form force assembly
    in ONE kernel per brick vs the two kernels.  The same arithmetic compiled into another kernel (other fma contractions): forces
    equal to a few units of the 2^-22 eV/A fixed point, energies and per-atom virials (the virial-only gather pass, which first
    has to bring the partial forces and the radial table back to HBM) to FP32 rounding; a 40-step run loop with list rebuilds
    stays on the same trajectory -- and the kernel itself against the oracle (check_force_parity's comparisons).
    The kernel is an experiment outside the default build (make -C gpumd_amd/csrc BRICK=1): skipped where the library lacks it."""
    probe = drv.engine(drv.model(H.golden("PbTe", "nep.txt")), 64)
    try:
        probe.set_brick_force(True)
    except Exception as exc:
        pytest.skip("libnepmi.so was built without the per-brick force kernel: %s" % exc)
    eng = P.check_force_parity(drv, "PbTe-ortho-big", lanes=1, win_static=True, force_form=1, brick=True)
    assert "one_kernel_per_brick" in eng.describe(), eng.describe()
    nep, (h, typ, x) = H.golden("PbTe", "nep.txt"), H.pbte_supercell((6, 6, 6), rattle=0.03, seed=17)
    mass = np.where(typ == 0, H.MASS["Te"], H.MASS["Pb"]).astype(np.float64)
    n = len(typ)
    vel = H.maxwell_velocities(mass, 2500.0, seed=4)
    m = drv.model(nep)
    out = []
    for brick in (True, False):
        eng = drv.engine(m, n)
        eng.set_win_lanes(1)
        eng.set_force_form(1)
        eng.set_brick_force(brick)
        _, pe0, f0, v0 = H.engine_force(drv, eng, h, typ, x)
        d_t, d_m, d_x, d_v = drv.dev(typ), drv.dev(mass), drv.dev(x), drv.dev(vel)
        d_pe, d_f, d_w = drv.zeros(n), drv.zeros(3 * n), drv.zeros(9 * n)
        eng.force_compute(h, d_t, d_x, d_pe, d_f, d_w)
        th = eng.run_nve(h, d_t, d_m, 2.0 / H.TIME_UNIT, 40, d_x, d_v, d_pe, d_f, d_w, thermo_every=10)
        out.append((pe0, f0, v0, drv.host(d_x), drv.host(d_v), drv.host(d_f), drv.host(d_pe), drv.host(d_w), np.asarray(th),
                    eng.describe(), eng.stats().num_rebuild))
    a, b = out
    assert "one_kernel_per_brick" in a[9] and "one_kernel_per_brick" not in b[9] and "lds_scatter" in b[9], (a[9], b[9])
    assert a[10] >= 2
    np.testing.assert_allclose(a[0], b[0], rtol=1e-6, atol=1e-6)      # energies
    assert np.abs(a[1] - b[1]).max() < 2e-6                           # forces: a few fixed-point units
    np.testing.assert_allclose(a[2], b[2], rtol=1e-5, atol=2e-5)      # per-atom virials
    assert np.abs(a[3] - b[3]).max() < 1e-6 and np.abs(a[4] - b[4]).max() < 1e-6 and np.abs(a[5] - b[5]).max() < 2e-4
    np.testing.assert_allclose(a[8], b[8], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("name", ["PbTe-A", "C-2022"])
def test_force_parity_with_pair_records(drv, name):
    """Tile mode 1: LDS-window radial pass that writes pair records + the record-reading force assembly."""
    P.check_force_parity(drv, name, check_lists=False, tiles=1)


def test_lds_window_pass_is_used(drv):
    eng = P.check_force_parity(drv, "PbTe-A", check_lists=False)
    assert eng.stats().radial_tiles >= 1


def test_invariances(drv):
    P.check_translation_and_wrap(drv)


@pytest.mark.parametrize("name", ["PbTe-A", "C-2022", "Si-5body"])
def test_rotation_permutation_and_finite_differences(drv, name):
    P.check_rotation_permutation_and_finite_differences(drv, name)


def test_average_of_two_potentials(drv):
    P.check_average_of_two_potentials(drv)


def test_nve_run(drv):
    P.check_nve_against_oracle(drv)


def test_streaming_ops(drv):
    P.check_streaming_ops(drv)


def test_unwrapped_positions(drv):
    P.check_unwrapped_positions(drv)


def test_nvt_berendsen(drv):
    P.check_nvt_berendsen(drv)


def test_nvt_nose_hoover_chain(drv):
    P.check_nvt_nhc(drv)


def test_angular_sums_stored_or_recomputed(drv):
    P.check_angular_recompute(drv)


def test_nvt_bussi_donadio_parrinello(drv):
    P.check_nvt_bdp(drv)


def test_small_box_branch(drv):
    P.check_small_box(drv)


def test_boundary_conditions_and_degenerate_inputs(drv):
    P.check_boundary_conditions(drv)


def test_error_paths(drv):
    P.check_error_paths(drv)


def test_native_library_is_loaded(drv):
    """The HIP extension is the thing that ran (no fallback): it is mapped into this process."""
    with open("/proc/self/maps") as f:
        assert "libnepmi.so" in f.read()


def test_larger_system_vs_oracle(drv):
    """PbTe 3x3x3 (6,750 atoms): still seconds for the oracle."""
    nep = H.golden("PbTe", "nep.txt")
    h, typ, x = H.pbte_supercell((3, 3, 3), seed=77)
    n = len(typ)
    pe64, f64, v64 = H.Oracle(nep).compute(typ, h, x, precision=64, path=0)
    eng = drv.engine(drv.model(nep), n)
    _, pe, f, v = H.engine_force(drv, eng, h, typ, x)
    np.testing.assert_allclose(pe.sum(), pe64.sum(), rtol=1e-5)
    assert np.all(np.abs(f - f64) <= 1e-4 * np.abs(f64) + 3e-5)


def test_full_size_properties(drv):
    """PbTe 1,024,000 atoms (BASELINE config 3, replicate 16 16 16): properties that do not need
    the oracle at this size."""
    torch = drv.torch
    nep = H.golden("PbTe", "nep.txt")
    h1, typ1, x1 = H.pbte_supercell((2, 2, 2), rattle=0.03, seed=123)   # 2000-atom rattled block
    # tile the rattled 2000-atom block 8x8x8 -> 1,024,000 atoms: every image of an atom sees the
    # same environment, so per-atom results must repeat with period 2000 (idempotence/periodicity)
    n1 = len(typ1)
    h, typ, pos = H.replicate(h1, typ1, x1.reshape(3, n1).T, (8, 8, 8))
    n = len(typ)
    assert n == 1024000
    x = H.soa(pos)
    eng = drv.engine(drv.model(nep), n)
    xw, pe, f, v = H.engine_force(drv, eng, h, typ.astype(np.int32), x)
    F = f.reshape(3, n)
    # (1) Newton's third law over the whole periodic system
    assert np.abs(F.sum(axis=1)).max() < 0.05
    # (2) periodic images agree.  Pairs that cross the periodic boundary go through the FP32
    # minimum-image round trip of a ~300 A difference (apply_mic, box.cuh:84-129: ulp 3e-5 A), so
    # boundary blocks carry ~1e-4 eV/A noise in the reference algorithm itself; interior blocks do not.
    blocks = F.reshape(3, 512, n1)
    inner = 3 * 64 + 4 * 8 + 3  # block (3,4,3) of the 8x8x8 tiling (replicate order i,j,k)
    assert np.abs(blocks - blocks[:, inner:inner + 1, :]).max() < 5e-4
    interior = [i * 64 + j * 8 + k for i in range(2, 6) for j in range(2, 6) for k in range(2, 6)]
    assert np.abs(blocks[:, interior, :] - blocks[:, inner:inner + 1, :]).max() < 3e-5
    pb = pe.reshape(512, n1)
    assert np.abs(pb - pb[inner:inner + 1]).max() < 5e-5
    # (3) and an interior block equals the oracle on the small periodic cell
    pe64, f64, _ = H.Oracle(nep).compute(typ1, h1, x1, precision=64, path=0)
    assert np.all(np.abs(blocks[:, inner, :].reshape(-1) - f64) <= 1e-4 * np.abs(f64) + 3e-5)
    np.testing.assert_allclose(pb[inner], pe64, rtol=1e-5, atol=2e-5)
    # (4) a second call on the same positions is bit-identical (deterministic, no atomics in sums)
    _, pe2, f2, v2 = H.engine_force(drv, eng, h, typ.astype(np.int32), x)
    assert np.array_equal(f2, f) and np.array_equal(pe2, pe) and np.array_equal(v2, v)
    st = eng.stats(True)
    assert st.num_rebuild == 1 and st.num_compute == 2


def test_long_nve_run_keeps_lists_and_energy(drv):
    """600 steps of 250,000 thermalising PbTe atoms: several Verlet rebuilds, no list-capacity overflow
    (the rattled crystal puts up to 21 atoms inside rc_a + skin), total energy conserved."""
    nep = H.golden("PbTe", "nep.txt")
    h, typ, x = H.pbte_supercell((10, 10, 10), rattle=0.02, seed=77)
    n = len(typ)
    model = drv.model(nep)
    mass = np.where(typ == 0, H.MASS["Te"], H.MASS["Pb"]).astype(np.float64)
    vel = H.maxwell_velocities(mass, 300.0, seed=9)
    eng = drv.engine(model, n)
    d_t, d_m, d_x, d_v = drv.dev(typ), drv.dev(mass), drv.dev(x), drv.dev(vel)
    d_pe, d_f, d_w = drv.zeros(n), drv.zeros(3 * n), drv.zeros(9 * n)
    eng.force_compute(h, d_t, d_x, d_pe, d_f, d_w)
    th = eng.run_nve(h, d_t, d_m, 1.0 / H.TIME_UNIT, 600, d_x, d_v, d_pe, d_f, d_w, thermo_every=100)
    st = eng.stats(with_lists=True)
    assert st.num_rebuild >= 4
    e = (th[:, 1] + 1.5 * n * H.K_B * th[:, 0]) / n
    assert np.abs(e - e[0]).max() < 5e-6, e          # eV per atom
    assert 300.0 < th[-1, 0] < 900.0  # model.xyz is a hot snapshot: potential energy flows into kinetic


def test_angular_list_overflow_is_an_error_not_a_fault(drv):
    """A run hot enough to put more than MN_angular atoms inside rc_a must end with the capacity error
    (nep.txt's cutoff line decides the capacity, as in the reference), never with a memory fault."""
    from gpumd_amd import NepmiError
    nep = H.golden("PbTe", "nep.txt")
    h, typ, x = H.pbte_supercell((4, 4, 4), rattle=0.02, seed=5)
    n = len(typ)
    mass = np.where(typ == 0, H.MASS["Te"], H.MASS["Pb"]).astype(np.float64)
    vel = H.maxwell_velocities(mass, 2500.0, seed=3)
    eng = drv.engine(drv.model(nep), n)
    d_t, d_m, d_x, d_v = drv.dev(typ), drv.dev(mass), drv.dev(x), drv.dev(vel)
    d_pe, d_f, d_w = drv.zeros(n), drv.zeros(3 * n), drv.zeros(9 * n)
    eng.force_compute(h, d_t, d_x, d_pe, d_f, d_w)
    with pytest.raises(NepmiError, match="capacity"):
        eng.run_nve(h, d_t, d_m, 1.0 / H.TIME_UNIT, 1500, d_x, d_v, d_pe, d_f, d_w)
