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
