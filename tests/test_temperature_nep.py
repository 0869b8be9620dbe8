"""Temperature-dependent NEP (nep4[_zbl]_temperature, src/force/nep.cu:125-130, :1392-1856): the ANN has one more input,
q[dim] = temperature * q_scaler[dim] (nep.cu:1483-1486), the temperature Force::compute hands to
NEP::compute(temperature, ...) (force.cu:516-525).  No shipped model is of this type and NEP_CPU does not carry it, so:

  * the oracle's handling is pinned against the reference's NEP_CPU through the algebraic identity behind it: a
    temperature model at temperature T is the plain nep4 model whose hidden bias is b0[j] - w0[j][dim] * T * q_scaler[dim];
  * the engine (emulator on the CPU tier, libnepmi on the GPU tier) is held against the oracle at several temperatures,
    in every ANN form (fused into the descriptor kernel, matrix-core kernel, per-atom kernel, generic shape);
  * tests/test_ref_md_parity.py runs the same synthetic model through the reference's own gpumd (GPU tier) with a
    temperature ramp.
"""
import numpy as np
import pytest

import helpers as H

DIM, NNEU, T = 42, 30, 2  # the shipped PbTe nep4 file


def _pbte_lines():
    with open(H.golden("PbTe", "nep.txt")) as f:
        return [x for x in f.read().split("\n") if x.strip()]


def make_temperature_model(tmp_path, zbl=False, name="nep_temperature.txt"):
    """PbTe nep4 + one ANN input: per type the w0 block grows from [neuron][dim] to [neuron][dim + 1] (the new column
    drawn at random, of the size of the other weights), q_scaler gets one more entry (1/1000 per kelvin)."""
    L = _pbte_lines()
    head, par = L[:6], L[6:]
    rng = np.random.default_rng(11)
    per = (DIM + 2) * NNEU
    out = [("nep4_zbl_temperature" if zbl else "nep4_temperature") + " 2 Te Pb"]
    if zbl:
        out += ["zbl 1.0 2.5"]
    out += head[1:]
    for t in range(T):
        blk = par[t * per:(t + 1) * per]
        w0 = np.array([float(v) for v in blk[:DIM * NNEU]]).reshape(NNEU, DIM)
        wt = rng.normal(0.0, 0.5, NNEU)
        w0n = np.concatenate([w0, wt[:, None]], axis=1)
        out += ["%.9e" % v for v in w0n.reshape(-1)] + blk[DIM * NNEU:]
    rest = par[T * per:]           # b1, descriptor coefficients, q_scaler[dim]
    out += rest + ["1.0e-03"]      # q_scaler of the temperature input
    p = tmp_path / name
    p.write_text("\n".join(out) + "\n")
    return str(p), head


def folded_plain_model(tmp_path, temp_model_path, temperature, name="folded.txt"):
    """The plain nep4 model a temperature model reduces to at one temperature (bias folding, in double)."""
    with open(temp_model_path) as f:
        L = [x for x in f.read().split("\n") if x.strip()]
    nhead = 7 if L[0].startswith("nep4_zbl") else 6
    head, par = L[:nhead], [float(v) for v in L[nhead:]]
    per_t = (DIM + 3) * NNEU
    qs_t = par[-1]
    out = [head[0].replace("_temperature", "")] + head[1:]
    body = []
    for t in range(T):
        blk = par[t * per_t:(t + 1) * per_t]
        w0 = np.array(blk[:(DIM + 1) * NNEU]).reshape(NNEU, DIM + 1)
        b0 = np.array(blk[(DIM + 1) * NNEU:(DIM + 2) * NNEU])
        w1 = blk[(DIM + 2) * NNEU:]
        body += list(w0[:, :DIM].reshape(-1)) + list(b0 - w0[:, DIM] * temperature * qs_t) + list(w1)
    body += par[T * per_t:-1]
    p = tmp_path / name
    p.write_text("\n".join(out + ["%.17e" % v for v in body]) + "\n")
    return str(p)


@pytest.mark.skipif(not H.ref_available(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("zbl", [False, True])
def test_oracle_temperature_model_against_nep_cpu_through_bias_folding(tmp_path, zbl):
    nep, _ = make_temperature_model(tmp_path, zbl)
    h, typ, x = H.pbte_supercell((2, 2, 2), seed=17)
    orc = H.Oracle(nep)
    assert orc.info.dim == DIM + 1  # annmb.dim of the reference counts the temperature input (nep.cu:321-325)
    e_prev = None
    for temperature in (0.0, 300.0, 925.5):
        orc.set_temperature(temperature)
        pe, f, v = orc.compute(typ, h, x, precision=64, path=0)
        pe_r, f_r, v_r = H.RefNepCpu(folded_plain_model(tmp_path, nep, temperature)).compute(typ, h, x)
        np.testing.assert_allclose(pe, pe_r, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(f, f_r, rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(v, v_r, rtol=1e-8, atol=1e-9)
        if e_prev is not None:
            assert abs(pe.sum() - e_prev) > 1e-3  # the input does something
        e_prev = pe.sum()


def _check_engine(drv, tmp_path, zbl, setup):
    nep, _ = make_temperature_model(tmp_path, zbl)
    h, typ, x = H.pbte_supercell((2, 2, 2), seed=23)
    n = len(typ)
    orc = H.Oracle(nep)
    model = drv.model(nep)
    assert model.info.model_type == 3 and model.info.dim == DIM
    eng = drv.engine(model, n)
    setup(eng)
    for temperature in (0.0, 300.0, 1200.0, 300.0):
        orc.set_temperature(temperature)
        pe64, f64, v64 = orc.compute(typ, h, x, precision=64, path=0)
        eng.set_temperature(temperature)
        _, pe, f, v = H.engine_force(drv, eng, h, typ, x)
        np.testing.assert_allclose(pe.sum(), pe64.sum(), rtol=1e-5)
        np.testing.assert_allclose(pe, pe64, rtol=1e-4, atol=2e-5)
        assert np.all(np.abs(f - f64) <= 1e-4 * np.abs(f64) + 3e-5)
        assert np.all(np.abs(v - v64) <= 1e-4 * np.abs(v64) + 1e-4)


SETUPS = [("default", lambda e: None), ("mfma", lambda e: e.set_mfma(2)), ("per-atom-ann", lambda e: e.set_mfma(0)),
          ("generic", lambda e: e.set_generic(True))]


@pytest.mark.parametrize("name,setup", SETUPS[:1] + SETUPS[3:])
@pytest.mark.parametrize("zbl", [False, True])
def test_temperature_model_on_emulator(tmp_path, zbl, name, setup):
    _check_engine(H.EmuDriver(), tmp_path, zbl, setup)


@pytest.mark.gpu
@pytest.mark.parametrize("name,setup", SETUPS)
@pytest.mark.parametrize("zbl", [False, True])
def test_temperature_model_on_gpu(tmp_path, zbl, name, setup):
    _check_engine(H.GpuDriver(), tmp_path, zbl, setup)


def test_plain_model_ignores_the_temperature(tmp_path):
    drv = H.EmuDriver()
    h, typ, x = H.pbte_supercell((2, 2, 2), seed=5)
    model = drv.model(H.golden("PbTe", "nep.txt"))
    assert model.info.model_type == 0
    eng = drv.engine(model, len(typ))
    _, pe0, f0, _ = H.engine_force(drv, eng, h, typ, x)
    eng.set_temperature(750.0)
    _, pe1, f1, _ = H.engine_force(drv, eng, h, typ, x)
    assert np.array_equal(pe0, pe1) and np.array_equal(f0, f1)
