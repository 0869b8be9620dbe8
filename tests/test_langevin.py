"""Langevin thermostat (`ensemble nvt_lan`, Ensemble_LAN, src/integrate/ensemble_lan.cu; kernels of
src/integrate/langevin_utilities.cuh).

GPU tier: nepmi_lan_half_step against the reference's OWN kernels (initialize_curand_states, gpu_langevin,
gpu_find_momentum, gpu_correct_momentum compiled for gfx950 from the header where it lies into
oracle/_ref/liblangevin_ref.so, oracle/ref_langevin_wrap.hip) with the same seed: the velocities must be equal bit for bit
over several half-steps (same XORWOW streams, same draw order, same summation order of the momentum).
Both tiers: the ensemble drives a system from 100 K to its 300 K target and keeps the total momentum at zero (the emulator
uses a different normal generator: statistics only)."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LAN = os.path.join(ROOT, "oracle", "_ref", "liblangevin_ref.so")


@pytest.mark.gpu
def test_half_step_equals_the_reference_kernels_bit_for_bit():
    if not os.path.exists(REF_LAN):
        pytest.skip("oracle/_ref/liblangevin_ref.so not built (needs /root/reference at build time)")
    import torch
    drv = H.GpuDriver()
    ref = C.CDLL(REF_LAN)
    ref.ref_lan_init.argtypes = [C.c_void_p, C.c_int, C.c_int]
    ref.ref_lan_half.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    n, seed, t_coup = 20011, 424242, 100.0
    rng = np.random.default_rng(3)
    mass = rng.choice([H.MASS["Te"], H.MASS["Pb"]], n)
    vel0 = H.maxwell_velocities(mass, 150.0, seed=5)
    dev = torch.device("cuda:0")
    states = torch.zeros(ref.ref_lan_state_bytes() * n, dtype=torch.uint8, device=dev)
    t_m = torch.from_numpy(mass).to(dev)
    v_ref = torch.from_numpy(vel0.copy()).to(dev)
    v_mi = torch.from_numpy(vel0.copy()).to(dev)
    assert ref.ref_lan_init(states.data_ptr(), n, seed) == 0
    eng = drv.engine(drv.model(H.golden("PbTe", "nep.txt")), n)
    eng.lan_seed(seed)
    for k, temperature in enumerate((300.0, 300.0, 450.0, 450.0, 80.0)):
        c1 = np.exp(-0.5 / t_coup)
        c2 = np.sqrt((1.0 - c1 * c1) * H.K_B * temperature)
        assert ref.ref_lan_half(states.data_ptr(), n, float(c1), float(c2), t_m.data_ptr(), v_ref.data_ptr()) == 0
        eng.lan_half_step(temperature, t_coup, t_m, v_mi)
        torch.cuda.synchronize()
        a, b = v_ref.cpu().numpy(), v_mi.cpu().numpy()
        assert np.array_equal(a, b), (k, np.abs(a - b).max())
    p = (np.tile(mass, 3) * v_mi.cpu().numpy()).reshape(3, n).sum(axis=1)
    assert np.abs(p).max() < 1e-9 * np.abs(mass * np.abs(v_mi.cpu().numpy().reshape(3, n)).max()).sum()


def _relaxes_to_target(drv, reps=(3, 3, 3)):
    h, typ, x = H.pbte_supercell(reps, seed=4)
    n = len(typ)
    mass = np.where(typ == 0, H.MASS["Te"], H.MASS["Pb"]).astype(np.float64)
    vel = H.maxwell_velocities(mass, 100.0, seed=9)
    eng = drv.engine(drv.model(H.golden("PbTe", "nep.txt")), n)
    d_t, d_m, d_x, d_v = drv.dev(typ), drv.dev(mass), drv.dev(x), drv.dev(vel)
    pe, f, w = drv.zeros(n), drv.zeros(3 * n), drv.zeros(9 * n)
    eng.force_compute(h, d_t, d_x, pe, f, w, n=n)
    eng.lan_seed(7)
    th = eng.run_nvt_lan(h, d_t, d_m, 1.0 / H.TIME_UNIT, 120, 300.0, 300.0, 10.0, d_x, d_v, pe, f, w, thermo_every=20)
    assert np.isfinite(th).all()
    # tau = 10 steps: after 120 steps the kinetic temperature fluctuates around the target (N = 6750: sigma ~ 3.5 K, the
    # emulator tier's N = 2000: 6.5 K; the hot model.xyz snapshot keeps feeding potential energy in, so allow a one-sided margin)
    assert 270.0 < th[-1, 0] < 360.0, th[:, 0]
    v = drv.host(d_v).reshape(3, n)
    assert np.abs((v * mass[None, :]).sum(axis=1)).max() < 1e-8 * np.abs(v * mass[None, :]).sum()


def _bao_relaxes(drv, reps=(3, 3, 3)):
    """`ensemble nvt_bao` (Ensemble_BAO: B A O A, force, B): thermostats to the target and conserves nothing it should
    not; with T_coup -> infinity (no noise, c1 = 1) it is velocity Verlet, i.e. equal to the NVE run step for step."""
    h, typ, x = H.pbte_supercell(reps, seed=4)
    n = len(typ)
    mass = np.where(typ == 0, H.MASS["Te"], H.MASS["Pb"]).astype(np.float64)
    vel = H.maxwell_velocities(mass, 100.0, seed=9)
    eng = drv.engine(drv.model(H.golden("PbTe", "nep.txt")), n)

    def fresh():
        d = [drv.dev(a) for a in (typ, mass, x.copy(), vel.copy())]
        out = [drv.zeros(n), drv.zeros(3 * n), drv.zeros(9 * n)]
        eng.force_compute(h, d[0], d[2], *out, n=n)
        return d, out
    (d_t, d_m, d_x, d_v), (pe, f, w) = fresh()
    eng.lan_seed(11)
    th = eng.run_nvt_bao(h, d_t, d_m, 1.0 / H.TIME_UNIT, 120, 300.0, 300.0, 10.0, d_x, d_v, pe, f, w, thermo_every=20)
    assert np.isfinite(th).all() and 270.0 < th[-1, 0] < 360.0, th[:, 0]
    # the deterministic limit: B A A B with c1 = exp(-1e-300) = 1, c2 = 0 is velocity Verlet
    (d_t, d_m, d_x, d_v), (pe, f, w) = fresh()
    th_bao = eng.run_nvt_bao(h, d_t, d_m, 1.0 / H.TIME_UNIT, 10, 300.0, 300.0, 1e300, d_x, d_v, pe, f, w, thermo_every=5)
    x_bao = drv.host(d_x)
    (d_t, d_m, d_x, d_v), (pe, f, w) = fresh()
    th_nve = eng.run_nve(h, d_t, d_m, 1.0 / H.TIME_UNIT, 10, d_x, d_v, pe, f, w, thermo_every=5)
    np.testing.assert_allclose(th_bao[:, :3], th_nve[:, :3], rtol=2e-6)
    assert np.abs(x_bao - drv.host(d_x)).max() < 1e-6


def test_nvt_bao_on_emulator():
    _bao_relaxes(H.EmuDriver(), reps=(2, 2, 2))  # (the host loop is slow: 2,000 atoms here, 6,750 on the GPU tier)


@pytest.mark.gpu
def test_nvt_bao_on_gpu():
    _bao_relaxes(H.GpuDriver())


def test_nvt_lan_relaxes_to_the_target_on_emulator():
    _relaxes_to_target(H.EmuDriver(), reps=(2, 2, 2))


@pytest.mark.gpu
def test_nvt_lan_relaxes_to_the_target_on_gpu():
    _relaxes_to_target(H.GpuDriver())


def _resident_loop_equals_stepwise(drv, reps=(3, 3, 3)):
    """nepmi_run_nvt_lan runs device-resident (state in the engine's internal order, steps enqueued speculatively, rebuilds
    inside the run): the generator states stay in the caller's atom order and the momentum sums are formed in the caller's
    order, so it must reproduce the stepwise sequence of the per-call entry points -- lan_half_step, vv_step1, Force::compute (wrap,
    zero, force), vv_step2, lan_half_step -- BIT FOR BIT in the velocities and positions."""
    h, typ, x = H.pbte_supercell(reps, seed=4)
    n = len(typ)
    mass = np.where(typ == 0, H.MASS["Te"], H.MASS["Pb"]).astype(np.float64)
    vel = H.maxwell_velocities(mass, 1500.0, seed=9)  # hot: list rebuilds inside the 40 steps
    dt, nsteps, t_coup = 2.0 / H.TIME_UNIT, 40, 50.0
    model = drv.model(H.golden("PbTe", "nep.txt"))

    def start(eng):
        d = [drv.dev(a) for a in (typ, mass, x.copy(), vel.copy())]
        out = [drv.zeros(n), drv.zeros(3 * n), drv.zeros(9 * n)]
        eng.force_compute(h, d[0], d[2], *out, n=n)
        eng.lan_seed(2024)
        return d, out
    eng_a = drv.engine(model, n)
    (d_t, d_m, d_x, d_v), (pe, f, w) = start(eng_a)
    eng_a.run_nvt_lan(h, d_t, d_m, dt, nsteps, 600.0, 300.0, t_coup, d_x, d_v, pe, f, w)
    assert eng_a.stats().num_rebuild >= 2
    xa, va = drv.host(d_x), drv.host(d_v)
    eng_b = drv.engine(model, n)
    (d_t, d_m, d_x, d_v), (pe, f, w) = start(eng_b)
    for s in range(nsteps):
        target = 600.0 + (300.0 - 600.0) * (s / nsteps)
        eng_b.lan_half_step(target, t_coup, d_m, d_v)
        eng_b.vv_step1(dt, d_m, f, d_x, d_v)
        eng_b.force_compute(h, d_t, d_x, pe, f, w, n=n)  # Force::compute: wrap, zero, compute
        eng_b.vv_step2(dt, d_m, f, d_v)
        eng_b.lan_half_step(target, t_coup, d_m, d_v)
    xb, vb = drv.host(d_x), drv.host(d_v)
    # the force kernels sum in the engine's internal order in both runs, so the forces -- and with them everything -- agree
    # to the last bit as long as both runs rebuild their lists at the same steps (same skin policy on the same positions)
    assert np.array_equal(va, vb), np.abs(va - vb).max()
    assert np.array_equal(xa, xb), np.abs(xa - xb).max()


def test_resident_nvt_lan_equals_the_stepwise_sequence_on_emulator():
    _resident_loop_equals_stepwise(H.EmuDriver(), reps=(2, 2, 2))


@pytest.mark.gpu
def test_resident_nvt_lan_equals_the_stepwise_sequence_on_gpu():
    _resident_loop_equals_stepwise(H.GpuDriver())


def _resident_bao_equals_stepwise(drv, reps=(3, 3, 3)):
    """nepmi_run_nvt_bao device-resident (B A O A on the internal-order state, speculative enqueue, rebuilds inside) against the
    stepwise sequence of the same engine (nepmi_engine_set_stepwise_loops): bit for bit."""
    h, typ, x = H.pbte_supercell(reps, seed=4)
    n = len(typ)
    mass = np.where(typ == 0, H.MASS["Te"], H.MASS["Pb"]).astype(np.float64)
    vel = H.maxwell_velocities(mass, 1500.0, seed=9)
    model = drv.model(H.golden("PbTe", "nep.txt"))
    out = []
    for stepwise in (False, True):
        eng = drv.engine(model, n)
        eng.set_stepwise_loops(stepwise)
        d_t, d_m, d_x, d_v = [drv.dev(a) for a in (typ, mass, x.copy(), vel.copy())]
        pe, f, w = drv.zeros(n), drv.zeros(3 * n), drv.zeros(9 * n)
        eng.force_compute(h, d_t, d_x, pe, f, w, n=n)
        eng.lan_seed(77)
        th = eng.run_nvt_bao(h, d_t, d_m, 2.0 / H.TIME_UNIT, 40, 600.0, 600.0, 50.0, d_x, d_v, pe, f, w, thermo_every=10)
        out.append((drv.host(d_x), drv.host(d_v), th, eng.stats().num_rebuild))
    assert out[0][3] >= 2 and out[0][3] == out[1][3]
    assert np.array_equal(out[0][1], out[1][1]), np.abs(out[0][1] - out[1][1]).max()
    assert np.array_equal(out[0][0], out[1][0]), np.abs(out[0][0] - out[1][0]).max()
    np.testing.assert_allclose(out[0][2], out[1][2], rtol=1e-12)


def test_resident_nvt_bao_equals_the_stepwise_sequence_on_emulator():
    _resident_bao_equals_stepwise(H.EmuDriver(), reps=(2, 2, 2))


@pytest.mark.gpu
def test_resident_nvt_bao_equals_the_stepwise_sequence_on_gpu():
    _resident_bao_equals_stepwise(H.GpuDriver())
