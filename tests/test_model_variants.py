"""Model-format variants the shipped fixtures do not cover, synthesised from the PbTe nep4 file:
nep5 (per-type output bias), nep3 with two types (one shared network), per-type cutoffs.  The
engine (emulator on the CPU tier, library on the GPU tier) against the oracle, and the oracle
against the reference's NEP_CPU where NEP_CPU supports the variant."""
import numpy as np
import pytest

import helpers as H


def _pbte_lines():
    with open(H.golden("PbTe", "nep.txt")) as f:
        return f.read().split("\n")


def _write(tmp_path, name, lines):
    p = tmp_path / name
    p.write_text("\n".join(lines))
    return str(p)


def make_nep5(tmp_path):
    L = _pbte_lines()
    head, par = L[:6], L[6:]
    dim, nneu = 42, 30
    per = (dim + 2) * nneu
    out = ["nep5 2 Te Pb"] + head[1:]
    out += par[:per] + ["0.0321"] + par[per:2 * per] + ["-0.0456"] + par[2 * per:]
    return _write(tmp_path, "nep5.txt", out)


def make_nep3(tmp_path):
    L = _pbte_lines()
    head, par = L[:6], L[6:]
    per = (42 + 2) * 30
    out = ["nep3 2 Te Pb"] + head[1:] + par[:per] + par[2 * per:]   # one ANN block + b1 + c + q_scaler
    return _write(tmp_path, "nep3.txt", out)


def make_per_type_cutoff(tmp_path):
    L = _pbte_lines()
    out = [L[0], "cutoff 8 4 7.2 3.6 73 8"] + L[2:]
    return _write(tmp_path, "pertype.txt", out)


def make_flexible_zbl(tmp_path):
    """nep4_zbl with `zbl 0 0`: the flexible ZBL (nep.cu:176-178, :925-932) -- ten parameters per type pair
    (rc_inner, rc_outer, then four (a, b) pairs of the screening function) behind q_scaler.  The pair is brought close
    enough (rattled PbTe has 2.9-3.3 A bonds; outer cutoffs 3.4-3.8 A) for the repulsion to act."""
    L = [x for x in _pbte_lines() if x.strip()]
    out = ["nep4_zbl 2 Te Pb", "zbl 0 0"] + L[1:]
    rng = np.random.default_rng(3)
    for pair in range(3):  # Te-Te, Te-Pb, Pb-Pb
        a = rng.uniform(0.1, 0.5, 4)
        a /= a.sum()
        b = np.array([3.2, 0.94, 0.40, 0.20]) * rng.uniform(0.9, 1.1, 4)
        vals = [1.2 + 0.1 * pair, 3.4 + 0.2 * pair]
        for k in range(4):
            vals += [a[k], b[k]]
        out += ["%.8e" % v for v in vals]
    return _write(tmp_path, "flexzbl.txt", out)


def make_typewise_zbl(tmp_path, factor=1.15, name="typewise.txt"):
    """`zbl rc_inner rc_outer factor`: the universal ZBL with a type-wise outer cutoff
    min((R_cov(Z1) + R_cov(Z2)) * factor, rc_outer) and inner cutoff 0 (nep.cu:179-186, :935-941)."""
    L = [x for x in _pbte_lines() if x.strip()]
    return _write(tmp_path, name, ["nep4_zbl 2 Te Pb", "zbl 1.0 3.6 %g" % factor] + L[1:])


@pytest.mark.skipif(not H.ref_available(), reason="oracle/_ref not built (no /root/reference here)")
def test_typewise_zbl_oracle_against_nep_cpu(tmp_path):
    """The vendored NEP_CPU takes rc_inner = rc_outer / 2 under a type-wise ZBL cutoff (nep.cpp:1397-1402) where the
    GPU code this engine follows takes 0 (nep.cu:935-941), so it cannot evaluate the variant itself -- but it can
    evaluate the universal ZBL that the type-wise rule reduces to: (1) a factor so large that every pair keeps
    rc_outer == `zbl 0 rc_outer`; (2) an all-Te structure, where the one pair cutoff is (2 R_cov(Te)) * factor
    == `zbl 0 <that>`.  The oracle must agree with NEP_CPU on both."""
    h, typ, x = H.pbte_supercell((2, 2, 2), seed=17)
    L = [ln for ln in _pbte_lines() if ln.strip()]

    def universal(rc_outer, name):
        return _write(tmp_path, name, ["nep4_zbl 2 Te Pb", "zbl 0.0 %.9g" % rc_outer] + L[1:])

    pe, f, _ = H.Oracle(make_typewise_zbl(tmp_path, 100.0, "tw_big.txt")).compute(typ, h, x, precision=64, path=0)
    pe_r, f_r, _ = H.RefNepCpu(universal(3.6, "uni36.txt")).compute(typ, h, x)
    np.testing.assert_allclose(f, f_r, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(pe, pe_r, rtol=1e-10, atol=1e-10)
    te = np.zeros_like(typ)
    r_te = 1.64  # covalent radius of Te (Z = 52) in the reference's table, nep_utilities.cuh:143-154
    pe, f, _ = H.Oracle(make_typewise_zbl(tmp_path, 1.05, "tw_105.txt")).compute(te, h, x, precision=64, path=0)
    pe_r, f_r, _ = H.RefNepCpu(universal(np.float32(2 * np.float32(r_te)) * np.float32(1.05), "uni_te.txt")).compute(te, h, x)
    np.testing.assert_allclose(f, f_r, rtol=1e-6, atol=1e-7)   # the cutoff itself is an FP32 product in the reference
    np.testing.assert_allclose(pe, pe_r, rtol=1e-7, atol=1e-7)
    # and the repulsion really acts inside 2 R_cov(Te) * 1.05 = 3.44 A
    pe0, _, _ = H.Oracle(make_typewise_zbl(tmp_path, 1.0e-3, "tw_off.txt")).compute(te, h, x, precision=64, path=0)
    assert abs(pe.sum() - pe0.sum()) > 1e-3


def make_extra_rows(flags, l_max=4):
    """l_max 4 <222> <1111> <112> <123> <233> <134>: the PbTe descriptor with the optional 4-body rows switched on
    (nep.cu:262-312).  No shipped model has them, so the ANN is seeded random (the descriptor coefficients and the
    scalers of the first rows stay those of the PbTe file)."""
    def maker(tmp_path):
        L = _pbte_lines()
        head, par = L[:6], [x for x in L[6:] if x.strip()]
        old_dim, nneu, nA1 = 42, 30, 7
        num_L = l_max + sum(flags)
        dim = 7 + nA1 * num_L
        n_c = len(par) - (2 * (old_dim + 2) * nneu + 1) - old_dim
        c = par[2 * (old_dim + 2) * nneu + 1: 2 * (old_dim + 2) * nneu + 1 + n_c]
        scaler = par[-old_dim:]
        rng = np.random.default_rng(sum(b << k for k, b in enumerate(flags)))
        ann = []
        for _ in range(2):
            ann += list(rng.normal(0, 0.4, dim * nneu)) + list(rng.normal(0, 0.3, nneu)) + list(rng.normal(0, 0.5, nneu))
        ann.append(-3.21)
        rows = scaler[:7 + l_max * nA1] + ["%.8e" % v for v in rng.uniform(0.5, 3.0, nA1 * (num_L - l_max))]
        # the 222 flag is written the way nep.txt files carry it (l_max_4body = 2; NEP_CPU tests == 2, nep.cu != 0)
        toks = [str(2 * b if k == 0 else b) for k, b in enumerate(flags)]
        out = head[:4] + ["l_max %d " % l_max + " ".join(toks), "ANN %d 0" % nneu]
        out += ["%.8e" % v for v in ann] + c + rows
        return _write(tmp_path, "extra_%d_%s.txt" % (l_max, "".join(str(b) for b in flags)), out)
    return maker


def _oracle_finite_differences(orc, h, typ, x):
    """F = -dE/dx of the oracle itself (FP64, 1e-4 A central differences): the only end-to-end check available for
    rows that NEP_CPU does not carry; the rows themselves are pinned in test_oracle_golden.py."""
    pe, f, _ = orc.compute(typ, h, x, precision=64, path=0)
    n = len(typ)
    rng = np.random.default_rng(1)
    for k in rng.integers(0, 3 * n, 4):
        xp, xm = x.copy(), x.copy()
        xp[k] += 1e-4
        xm[k] -= 1e-4
        num = -(orc.compute(typ, h, xp, precision=64, path=0)[0].sum() -
                orc.compute(typ, h, xm, precision=64, path=0)[0].sum()) / 2e-4
        assert abs(num - f[k]) < 1e-6 * (1 + abs(f[k])), (k, num, f[k])


def _check(drv, nep, ref_cpu):
    h, typ, x = H.pbte_supercell((2, 2, 2), seed=17)
    n = len(typ)
    orc = H.Oracle(nep)
    pe64, f64, v64 = orc.compute(typ, h, x, precision=64, path=0)
    if "extra_" in nep or "typewise" in nep:
        _oracle_finite_differences(orc, h, typ, x)
    if ref_cpu and H.ref_available():
        pe_r, f_r, v_r = H.RefNepCpu(nep).compute(typ, h, x)
        np.testing.assert_allclose(f64, f_r, rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(pe64, pe_r, rtol=1e-10, atol=1e-10)
    eng = drv.engine(drv.model(nep), n)
    _, pe, f, v = H.engine_force(drv, eng, h, typ, x)
    np.testing.assert_allclose(pe.sum(), pe64.sum(), rtol=1e-5)
    assert np.all(np.abs(f - f64) <= 1e-4 * np.abs(f64) + 3e-5)
    assert np.all(np.abs(v - v64) <= 1e-4 * np.abs(v64) + 1e-4)
    L = orc.lists(typ, h, x, path=0)
    for which, key in ((0, "radial"), (1, "angular")):
        onn, onl = L[key]
        _, nn, nl = H.engine_lists(drv, eng, n, which, ld=int(onn.max()) + 2)
        H.assert_lists_equal(nn, nl, onn, onl)


VARIANTS = [("flexible-zbl", make_flexible_zbl, True), ("typewise-zbl", make_typewise_zbl, False),
            ("nep5", make_nep5, True), ("nep3-2types", make_nep3, True), ("per-type-cutoff", make_per_type_cutoff, False),
            # the optional 4-body rows 112 / 123 / 233 / 134 (flags: 222 1111 112 123 233 134)
            ("rows-all", make_extra_rows([1, 1, 1, 1, 1, 1]), False), ("rows-112", make_extra_rows([1, 0, 1, 0, 0, 0]), False),
            ("rows-123-233", make_extra_rows([0, 0, 0, 1, 1, 0]), False), ("rows-134", make_extra_rows([0, 1, 0, 0, 0, 1]), False),
            # l_max_3body below 4 (generic shape).  NEP_CPU carries these too -- except a 1111 row without a 222 row, which
            # it reads as the 222 row (nep.cpp:642-660) -- so the oracle is held against it where it can be
            ("lmax-2-222-1111", make_extra_rows([1, 1], 2), True), ("lmax-1-1111", make_extra_rows([0, 1], 1), False),
            ("lmax-3", make_extra_rows([0, 0], 3), True), ("lmax-3-222-112-233", make_extra_rows([1, 0, 1, 0, 1], 3), False)]


@pytest.mark.parametrize("name,maker,ref_cpu", VARIANTS)
def test_variant_on_emulator(tmp_path, name, maker, ref_cpu):
    _check(H.EmuDriver(), maker(tmp_path), ref_cpu)


@pytest.mark.gpu
@pytest.mark.parametrize("name,maker,ref_cpu", VARIANTS)
def test_variant_on_gpu(tmp_path, name, maker, ref_cpu):
    _check(H.GpuDriver(), maker(tmp_path), ref_cpu)
