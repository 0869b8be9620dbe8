"""Tersoff-1989 (BASELINE config 2: Si, classical pair-force + Verlet path, FP64).

There is no force-level golden vector for this potential in the reference tree; the oracle
(oracle/tersoff_oracle.c, restating src/force/tersoff1989.cu) is pinned against the reference's OWN kernels
(find_force_tersoff_step1/2 and gpu_find_force_many_body compiled for the host by oracle/ref_tersoff_wrap.cpp) and
checked by finite differences, Newton's third law and the strain derivative of its own energy; the engine is then
compared with the oracle (FP64: agreement ~1e-10) -- CPU tier through the kernel emulator, GPU tier on the device."""
import os
import subprocess

import numpy as np
import pytest

import helpers as H

POT = H.golden("Si", "Si_Tersoff_1989.txt")


def test_oracle_self_consistency():
    o = H.TersoffOracle(POT)
    h, typ, x = H.diamond((3, 3, 3), 5.432, rattle=0.08, seed=2)
    n = len(typ)
    pe, f, v, nn, nl = o.compute(typ, h, x, lists=True)
    assert nn.min() == 4 and nn.max() == 4                     # diamond: 4 bonds inside r2 = 3 A
    assert -4.7 < pe.sum() / n < -4.3                           # Si cohesive energy ~ -4.63 eV
    assert np.abs(f.reshape(3, n).sum(axis=1)).max() < 1e-12    # Newton's third law
    rng = np.random.default_rng(0)
    for _ in range(5):                                          # central differences of the energy
        i, d, eps = rng.integers(0, n), rng.integers(0, 3), 1e-5
        xp, xm = x.copy(), x.copy()
        xp[d * n + i] += eps
        xm[d * n + i] -= eps
        fd = -(o.compute(typ, h, xp)[0].sum() - o.compute(typ, h, xm)[0].sum()) / (2 * eps)
        assert abs(fd - f[d * n + i]) < 1e-6
    eps = 1e-6                                                  # virial = -dE/d(strain)

    def e_strain(e, a, b):
        S = np.eye(3)
        S[a, b] += e
        H2 = S @ np.asarray(h).reshape(3, 3)
        return o.compute(typ, H2.reshape(9), (S @ x.reshape(3, n)).reshape(-1))[0].sum()

    W = v.reshape(9, n).sum(axis=1)
    for (a, b), comp in (((0, 0), 0), ((1, 1), 1), ((2, 2), 2), ((0, 1), 3)):
        dE = (e_strain(eps, a, b) - e_strain(-eps, a, b)) / (2 * eps)
        assert abs(dE + W[comp]) < 1e-5 * max(1.0, abs(W[comp]))


def test_oracle_cell_candidates_equal_full_scan():
    """the oracle's O(N) candidate narrowing (used above 512 atoms) gives bit-identical lists and outputs to its
    defining all-pairs scan, orthogonal and triclinic"""
    o = H.TersoffOracle(POT)
    for tri in (False, True):
        h, typ, x = H.diamond((5, 4, 6), 5.432, rattle=0.08, seed=9)
        if tri:
            Hm = np.asarray(h).reshape(3, 3).copy()
            Hm[0, 1], Hm[0, 2], Hm[1, 2] = 0.9, -0.7, 1.1
            frac = np.linalg.solve(np.asarray(h).reshape(3, 3), x.reshape(3, -1))
            h, x = Hm.reshape(9), (Hm @ frac).reshape(-1)
        x = x + np.repeat(np.asarray(h).reshape(3, 3)[:, 0] * 2.0, len(typ))      # some atoms outside the cell
        o.L.terso_full_scan(1)
        try:
            a = o.compute(typ, h, x, lists=True)
        finally:
            o.L.terso_full_scan(0)
        b = o.compute(typ, h, x, lists=True)
        for u, v in zip(a, b):
            assert np.array_equal(u, v)


@pytest.mark.skipif(not os.path.exists(os.path.join(H.ROOT, "oracle", "_ref", "libtersoff_ref.so")),
                    reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("case", ["Si-ortho", "Si-triclinic", "two-types"])
def test_oracle_vs_reference_kernels(case, tmp_path):
    """oracle/tersoff_oracle.c == the reference's own Tersoff kernels (tersoff1989.cu:157-505, potential.cu:35-134)
    compiled for the host, on the same local neighbour list: energies, forces and all nine virial planes to 1e-11."""
    import ctypes as C
    pot = POT
    h, typ, x = H.diamond((3, 3, 4), 5.432, rattle=0.08, seed=12)
    n = len(typ)
    if case == "Si-triclinic":
        S = np.array([[1.0, 0.07, -0.04], [0.0, 1.0, 0.05], [0.0, 0.0, 1.0]])
        h = (S @ np.asarray(h).reshape(3, 3)).reshape(9)
        x = (S @ x.reshape(3, n)).reshape(-1)
    if case == "two-types":  # a synthetic SiC-like file: two parameter sets + the mixing factor chi
        lines = open(POT).read().split("\n")
        assert lines[0].split()[:2] == ["tersoff_1989", "1"]
        p0 = lines[1].split()
        p1 = [("%.10g" % (float(v) * f)) for v, f in zip(p0, (0.8, 0.9, 1.05, 1.1, 1.3, 0.95, 1.0, 1.0, 1.0, 0.95, 0.97))]
        pot = str(tmp_path / "two.txt")
        open(pot, "w").write("tersoff_1989 2 Si C\n%s\n%s\n0.98\n" % (" ".join(p0), " ".join(p1)))
        typ = (np.arange(n) % 2).astype(np.int32)
    o = H.TersoffOracle(pot)
    pe_o, f_o, v_o, nn, nl = o.compute(typ, h, x, lists=True)
    par = np.zeros(48)
    o.L.terso_params(o.h, H._p(par, H._dp))
    H3 = np.asarray(h, dtype=np.float64).reshape(3, 3)
    h18 = np.ascontiguousarray(np.concatenate([H3.reshape(9), np.linalg.inv(H3).reshape(9)]))
    ortho = int(np.count_nonzero(H3 - np.diag(np.diag(H3))) == 0)
    L = C.CDLL(os.path.join(H.ROOT, "oracle", "_ref", "libtersoff_ref.so"))
    nl_full = np.ascontiguousarray(np.where(nl < 0, 0, nl).astype(np.int32))  # [slot][atom], stride n
    pe, f, v = np.zeros(n), np.zeros(3 * n), np.zeros(9 * n)
    pbc = np.array([1, 1, 1], dtype=np.int32)
    xw = H.oracle_apply_pbc(h, x)
    L.nepref_tersoff(C.c_int(n), H._p(h18, H._dp), H._p(pbc, H._ip), C.c_int(ortho), H._p(par, H._dp),
                     H._p(np.ascontiguousarray(nn.astype(np.int32)), H._ip), H._p(nl_full, H._ip),
                     H._p(np.ascontiguousarray(typ.astype(np.int32)), H._ip), H._p(np.ascontiguousarray(xw), H._dp),
                     H._p(pe, H._dp), H._p(f, H._dp), H._p(v, H._dp))
    pe_o, f_o, v_o, _, _ = o.compute(typ, h, xw, lists=True)
    assert nn.max() >= 4
    np.testing.assert_allclose(pe, pe_o, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(f, f_o, rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(v, v_o, rtol=1e-10, atol=1e-11)


def _check_engine(drv, cells=(4, 4, 5), nve_steps=40):
    o = H.TersoffOracle(POT)
    h, typ, x = H.diamond(cells, 5.432, rattle=0.06, seed=3)
    n = len(typ)
    pe_o, f_o, v_o, nn_o, nl_o = o.compute(typ, h, x, lists=True)
    model = drv.model(POT)
    assert model.symbols == ["Si"] and abs(model.info.rc_radial - 3.0) < 1e-12
    eng = drv.engine(model, n)
    xw, pe, f, v = H.engine_force(drv, eng, h, typ, x)
    assert np.array_equal(xw, H.oracle_apply_pbc(h, x))
    np.testing.assert_allclose(pe, pe_o, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(f, f_o, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(v, v_o, rtol=1e-9, atol=1e-9)
    mx, nn, nl = H.engine_lists(drv, eng, n, 0, ld=int(nn_o.max()) + 2)
    H.assert_lists_equal(nn, nl, nn_o, nl_o)                    # local list bit-exact
    # NVE: energy conservation (FP64 forces: tight) and a list rebuild on the way
    mass = np.full(n, 28.085)
    vel = H.maxwell_velocities(mass, 2500.0, seed=4)
    d_t, d_m, d_x, d_v = drv.dev(typ), drv.dev(mass), drv.dev(x), drv.dev(vel)
    d_pe, d_f, d_w = drv.zeros(n), drv.zeros(3 * n), drv.zeros(9 * n)
    eng = drv.engine(model, n)
    eng.force_compute(h, d_t, d_x, d_pe, d_f, d_w)
    th = eng.run_nve(h, d_t, d_m, 1.0 / H.TIME_UNIT, nve_steps, d_x, d_v, d_pe, d_f, d_w, thermo_every=1)
    etot = 1.5 * n * H.K_B * th[:, 0] + th[:, 1]
    assert np.abs(etot - etot[0]).max() < 1e-3 * n  # O(dt^2) fluctuation at 2500 K, dt = 1 fs
    # final forces of the trajectory still equal the oracle on the final positions
    pe_o2, f_o2, _ = o.compute(typ, h, drv.host(d_x))
    np.testing.assert_allclose(drv.host(d_f), f_o2, rtol=1e-8, atol=1e-8)
    return eng


def test_engine_logic_on_emulator():
    eng = _check_engine(H.EmuDriver(), cells=(3, 3, 4), nve_steps=25)
    assert eng.stats().num_rebuild >= 1


@pytest.mark.gpu
def test_engine_on_gpu():
    _check_engine(H.GpuDriver())


@pytest.mark.gpu
def test_config2_si_13824_nve(tmp_path):
    """BASELINE config 2 through gpumd-mi: Si 13,824 atoms (12x12x12 diamond cells), Tersoff NVE."""
    h, typ, x = H.diamond((12, 12, 12), 5.432, rattle=0.0, seed=1)
    n = len(typ)
    assert n == 13824
    pos = x.reshape(3, n).T
    with open(tmp_path / "model.xyz", "w") as f:
        f.write("%d\npbc=\"T T T\" Lattice=\"%.10f 0 0 0 %.10f 0 0 0 %.10f\" Properties=species:S:1:pos:R:3\n"
                % (n, h[0], h[4], h[8]))
        for p in pos:
            f.write("Si %.12f %.12f %.12f\n" % tuple(p))
    (tmp_path / "run.in").write_text("potential %s\nvelocity 300 seed 42\nensemble nve\ntime_step 1\n"
                                     "dump_thermo 100\nrun 1000\n" % POT)
    exe = os.path.join(H.ROOT, "gpumd_amd", "bin", "gpumd-mi")
    out = subprocess.run([exe], cwd=str(tmp_path), capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    th = np.loadtxt(tmp_path / "thermo.out")
    assert th.shape == (10, 18)
    etot = th[:, 1] + th[:, 2]
    assert np.abs(etot - etot[0]).max() < 5e-5 * n              # < 5e-5 eV/atom fluctuation over 1 ps
    assert 120.0 < th[-1, 0] < 180.0                             # equipartition from 300 K
    speed = [l for l in out.stdout.splitlines() if "atom*step/second" in l]
    assert speed, out.stdout
    print(speed[0])
