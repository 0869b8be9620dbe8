"""CPU tier: the kernel bodies + engine sequencing run by the host-loop emulator (tests/emu, test
infrastructure) against the oracle.  Validates the kernel LOGIC where no GPU exists; the same
cases run on the real library under -m gpu (test_gpu_parity.py)."""
import pytest

import helpers as H
import parity_cases as P


@pytest.fixture(scope="module")
def drv():
    return H.EmuDriver()


@pytest.mark.parametrize("name", list(P.MODELS))
def test_force_parity(drv, name):
    P.check_force_parity(drv, name)


@pytest.mark.parametrize("name", ["PbTe-A", "UNEP-v1", "C-2022"])
def test_force_parity_generic_shape(drv, name):
    P.check_force_parity(drv, name, generic=True, check_lists=False)


@pytest.mark.parametrize("name,forced", [("Si-3body", False), ("Si-4body", False), ("water-model", True), ("PbTe-A", True), ("C-2022", True)])
def test_force_parity_zero_padded_into_a_cover_shape(drv, name, forced, monkeypatch):
    """A model of a shape nobody compiled kernels for is zero-padded into a compiled COVER shape (nep_model.h: embed_model; here
    with NEPMI_JIT=2: existing cores only): zero coefficients on the padded radial functions and basis functions, zero ANN
    weights on the padded descriptor components -- the model's own energies, forces, virials, lists and (through the component
    map) descriptors, against the oracle with the suite's tolerances.  forced: a model of a compiled shape padded all the same."""
    monkeypatch.setenv("NEPMI_JIT", "2")
    if forced:
        monkeypatch.setenv("NEPMI_FORCE_COVER", "1")
    eng = P.check_force_parity(drv, name)
    assert "shape=cover(" in eng.describe() and "zero_padded_model" in eng.describe(), eng.describe()


@pytest.mark.parametrize("name", ["PbTe-A", "PbTe-ortho", "UNEP-v1"])
def test_force_parity_without_lds_window(drv, name):
    """the plain gather radial kernel (fallback of the LDS-window pass) gives the same answers"""
    P.check_force_parity(drv, name, tiles=False)


@pytest.mark.parametrize("static", [True, False])
@pytest.mark.parametrize("name", ["PbTe-A", "PbTe-B", "C-2022", "UNEP-v1", "UNEP-v1-big", "BaZrO3", "PbTe-3x3x3"])
def test_force_parity_window_layouts(drv, name, static):
    """One lane per atom on the static window layout (Verlet entries kept as LDS slots, four to a word; two-type models
    walk list B as two type-pure streams) and on the scanned layout: the same lists bit for bit."""
    eng = P.check_force_parity(drv, name, lanes=1, win_static=static)
    if name == "UNEP-v1-big":  # many types: the neighbour's half of a pair force comes from its radial Fp row (static layout)
        assert ("neighbour_half_from_fp_rows" in eng.describe()) == bool(static)


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
