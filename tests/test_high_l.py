"""l_max_3body = 5..8 (the reference carries L up to 8: NUM_OF_ABC = 80 sums per radial order,
src/utilities/nep_utilities.cuh:18, accumulate_s / find_q / accumulate_f12 :1436-1830).  No shipped model uses it, so the
models are synthesised from the PbTe file (random ANN, the file's descriptor coefficients):

  * the oracle (tables for l = 5..8 generated from their definition by gpumd_amd/csrc/tools/gen_highl_tables.py) is
    pinned against the reference's NEP_CPU, which carries every l;
  * the engine (emulator / GPU) against the oracle, lists included;
  * tests/test_ref_md_parity.py runs one of the models through the reference's own gpumd (GPU tier).
"""
import numpy as np
import pytest

import helpers as H


def _pbte_lines():
    with open(H.golden("PbTe", "nep.txt")) as f:
        return [x for x in f.read().split("\n") if x.strip()]


def make_high_l(tmp_path, l_max, flags=(1, 1), name=None):
    """`l_max <l_max> <222> <1111>` on the PbTe descriptor; rows beyond the file's get the scaler of its l = 4 row."""
    L = _pbte_lines()
    head, par = L[:6], L[6:]
    old_dim, nneu, nA1 = 42, 30, 7
    num_L = l_max + sum(flags)
    dim = 7 + nA1 * num_L
    n_c = len(par) - (2 * (old_dim + 2) * nneu + 1) - old_dim
    c = par[2 * (old_dim + 2) * nneu + 1: 2 * (old_dim + 2) * nneu + 1 + n_c]
    scaler = par[-old_dim:]
    rng = np.random.default_rng(100 * l_max + sum(b << k for k, b in enumerate(flags)))
    ann = []
    for _ in range(2):
        ann += list(rng.normal(0, 0.4, dim * nneu)) + list(rng.normal(0, 0.3, nneu)) + list(rng.normal(0, 0.5, nneu))
    ann.append(-1.23)
    row4 = scaler[7 + 3 * nA1: 7 + 4 * nA1]
    rows = scaler[:7 + 4 * nA1] + row4 * (l_max - 4) + ["%.8e" % v for v in rng.uniform(0.5, 3.0, nA1 * sum(flags))]
    toks = [str(2 * b if k == 0 else b) for k, b in enumerate(flags)]  # the 222 flag as nep.txt files carry it (2)
    out = head[:4] + ["l_max %d " % l_max + " ".join(toks), "ANN %d 0" % nneu] + ["%.8e" % v for v in ann] + c + rows
    p = tmp_path / (name or "high_l_%d_%s.txt" % (l_max, "".join(str(b) for b in flags)))
    p.write_text("\n".join(out) + "\n")
    return str(p)


CASES = [(5, (1, 1)), (6, (0, 0)), (7, (1, 0)), (8, (1, 1))]


@pytest.mark.skipif(not H.ref_available(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("l_max,flags", CASES)
def test_oracle_high_l_against_nep_cpu(tmp_path, l_max, flags):
    nep = make_high_l(tmp_path, l_max, flags)
    h, typ, x = H.pbte_supercell((2, 2, 2), seed=17)
    orc = H.Oracle(nep)
    assert orc.info.L_max == l_max and orc.info.dim == 7 + 7 * (l_max + sum(flags))
    pe, f, v = orc.compute(typ, h, x, precision=64, path=0)
    pe_r, f_r, v_r = H.RefNepCpu(nep).compute(typ, h, x)
    np.testing.assert_allclose(pe, pe_r, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(f, f_r, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(v, v_r, rtol=1e-9, atol=1e-10)
    # and the FP32 instantiation (what the engine is compared with) stays within FP32 noise of it
    pe32, f32, _ = orc.compute(typ, h, x, precision=32, path=0)
    assert np.all(np.abs(f32 - f) <= 1e-4 * np.abs(f) + 3e-5)


def _check_engine(drv, nep):
    h, typ, x = H.pbte_supercell((2, 2, 2), seed=29)
    n = len(typ)
    orc = H.Oracle(nep)
    pe64, f64, v64 = orc.compute(typ, h, x, precision=64, path=0)
    eng = drv.engine(drv.model(nep), n)
    _, pe, f, v = H.engine_force(drv, eng, h, typ, x)
    np.testing.assert_allclose(pe.sum(), pe64.sum(), rtol=1e-5)
    np.testing.assert_allclose(pe, pe64, rtol=1e-4, atol=3e-5)
    assert np.all(np.abs(f - f64) <= 1e-4 * np.abs(f64) + 3e-5)
    assert np.all(np.abs(v - v64) <= 1e-4 * np.abs(v64) + 1e-4)
    assert np.abs(f.reshape(3, n).sum(axis=1)).max() < 2e-3  # Newton's third law through the reverse slots
    L = orc.lists(typ, h, x, path=0)
    for which, key in ((0, "radial"), (1, "angular")):
        onn, onl = L[key]
        _, nn, nl = H.engine_lists(drv, eng, n, which, ld=int(onn.max()) + 2)
        H.assert_lists_equal(nn, nl, onn, onl)


@pytest.mark.parametrize("l_max,flags", CASES)
def test_high_l_on_emulator(tmp_path, l_max, flags):
    _check_engine(H.EmuDriver(), make_high_l(tmp_path, l_max, flags))


@pytest.mark.gpu
@pytest.mark.parametrize("l_max,flags", CASES)
def test_high_l_on_gpu(tmp_path, l_max, flags):
    _check_engine(H.GpuDriver(), make_high_l(tmp_path, l_max, flags))
