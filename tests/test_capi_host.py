"""CPU tier: the product library loads, exports every symbol include/nepmi.h declares, and its
host-only entry points (model parsing) work without a GPU.  No compute call is made here."""
import ctypes
import os
import re

import pytest

import helpers as H

LIB = os.path.join(H.ROOT, "gpumd_amd", "lib", "libnepmi.so")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    if not os.path.exists(LIB):
        g.build()
    from gpumd_amd import _capi
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    return _capi.bind(ctypes.CDLL(LIB))


def test_header_and_bindings_agree(lib):
    from gpumd_amd import _capi
    hdr = open(os.path.join(H.ROOT, "include", "nepmi.h")).read()
    declared = set(re.findall(r"\b(nepmi_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"nepmi_status"}
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.nepmi_version() == 200


@pytest.mark.parametrize("rel", ["PbTe/nep.txt", "PbTe/nep_B.txt", "C/nep.txt", "C/nep3.txt", "UNEP/nep.txt",
                                 "BaZrO3/nep.txt", "water/nep.txt"])
def test_model_parser_matches_oracle(lib, rel):
    from gpumd_amd.nep import Model
    path = H.golden(*rel.split("/"))
    m = Model(path, lib=lib)
    o = H.Oracle(path)
    for a, b in (("version", "version"), ("num_types", "num_types"), ("dim", "dim"), ("num_neurons", "num_neurons"),
                 ("n_max_radial", "n_max_radial"), ("n_max_angular", "n_max_angular"),
                 ("basis_size_radial", "basis_size_radial"), ("basis_size_angular", "basis_size_angular"),
                 ("L_max", "L_max"), ("has_q_222", "has_222"), ("has_q_1111", "has_1111"), ("num_L", "num_L"),
                 ("MN_radial", "MN_radial"), ("MN_angular", "MN_angular"), ("num_para", "num_para"),
                 ("zbl_enabled", "zbl_enabled"), ("zbl_flexible", "zbl_flexible")):
        va, vb = getattr(m.info, a), getattr(o.info, b)
        if a.startswith('has_'):
            va, vb = bool(va), bool(vb)
        assert va == vb, (a, va, vb)
    assert m.info.rc_radial == o.info.rc_radial_max and m.info.rc_angular == o.info.rc_angular_max
    assert m.symbols == o.symbols


def test_model_errors(lib):
    from gpumd_amd import NepmiError
    from gpumd_amd.nep import Model
    with pytest.raises(NepmiError):
        Model(H.golden("nope.txt"), lib=lib)
    with pytest.raises(NepmiError):
        Model(H.golden("PbTe", "run.in"), lib=lib)


def test_no_cpu_fallback():
    """Without a GPU the product refuses to construct an engine (fails loudly, no CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import gpumd_amd
    with pytest.raises(RuntimeError):
        gpumd_amd.NEP(H.golden("PbTe", "nep.txt"), 2000)
