"""Kernels compiled for a model's own shape (gpumd_amd/csrc/capi_jit.h): a model outside the library's five compiled shapes is
served by a JIT core -- the same sources compiled once more for that shape -- behind the same C ABI (capi_dispatch.inc: every
handle carries the function table of the library that made it).  The cores of two shipped potentials are built ahead of time
(__graft_entry__.build() -> gpumd_amd/lib/jit/, they travel to the GPU box); NEPMI_JIT=2 = cores that exist, never the compiler.
The reference runs any n_max / basis_size / neuron count through one code path (src/force/nep.cu:488-659, 774-861)."""
import glob
import os

import numpy as np
import pytest

import helpers as H
import parity_cases as P

JIT_DIR = os.path.join(H.ROOT, "gpumd_amd", "lib", "jit")
CASES = {"C-2024": "12_16_8_12_6_1", "C-2024-window": "12_16_8_12_6_1", "Si-5body": "10_10_10_10_6_1"}


def _core(shape):
    return glob.glob(os.path.join(JIT_DIR, "libnepmi_jit_%s_*.so" % shape))


@pytest.mark.parametrize("name", sorted(CASES))
def test_prebuilt_core_is_found_and_serves_the_model(name, monkeypatch, capfd):
    """no GPU needed: nepmi_model_load decides which library serves the model, loads the core (the name carries the hash of the
    sources: a stale core is not found) and every call on the handle is forwarded to it"""
    if not _core(CASES[name]):
        pytest.skip("no prebuilt JIT core for %s (python -c 'import __graft_entry__ as g; g.build()')" % name)
    import gpumd_amd
    nep = H.golden(*P.MODELS[name][0].split("/"))
    monkeypatch.setenv("NEPMI_JIT", "2")
    m = gpumd_amd.Model(nep)
    err = capfd.readouterr().err
    assert "no JIT core" not in err and "zero-padded" not in err, err
    monkeypatch.setenv("NEPMI_JIT", "0")
    m0 = gpumd_amd.Model(nep)
    import ctypes as C
    # the first word of a handle: the function table of the library that made it (capi_impl.h) -- two different libraries
    assert C.c_void_p.from_address(m.handle).value != C.c_void_p.from_address(m0.handle).value
    for f, _ in m.info._fields_:
        assert getattr(m.info, f) == getattr(m0.info, f), f
    m.close()
    m0.close()


def test_a_model_of_a_compiled_shape_stays_with_the_library(monkeypatch, capfd):
    import gpumd_amd
    monkeypatch.setenv("NEPMI_JIT", "2")
    m = gpumd_amd.Model(H.golden("PbTe", "nep.txt"))
    assert "nepmi:" not in capfd.readouterr().err
    m.close()


def test_without_a_core_a_cover_shape_serves_the_model(monkeypatch, capfd):
    """NEPMI_JIT=2 and no core for the shape (Si with the 4-body row only: 10,10,10,10,5;1): the model is zero-padded into the smallest
    compiled COVER shape that holds it (12,16,10,12, six rows; nep_model.h: embed_model) -- one line on stderr says so --, and
    nepmi_model_info keeps reporting the MODEL, not the kernels that serve it.  NEPMI_COVER=0: the run-time-shape kernels."""
    import gpumd_amd
    assert not _core("10_10_10_10_5_1")
    monkeypatch.setenv("NEPMI_JIT", "2")
    m = gpumd_amd.Model(H.golden("Si", "nep_4body.txt"))
    err = capfd.readouterr().err
    assert "zero-padded" in err and "n_max 12 10, basis_size 16 12" in err, err
    assert m.info.num_types == 1 and m.info.num_L == 5 and m.info.n_max_radial == 10 and m.info.basis_size_angular == 10
    assert m.info.dim == 11 + 11 * 5
    m.close()
    monkeypatch.setenv("NEPMI_COVER", "0")
    m = gpumd_amd.Model(H.golden("Si", "nep_4body.txt"))
    assert "run-time-shape kernels serve this model" in capfd.readouterr().err
    m.close()


def test_first_load_of_a_new_shape_compiles_its_core(monkeypatch, capfd, tmp_path):
    """the compiler path itself (no GPU needed, ~40 s of hipcc): NEPMI_JIT=1, an empty cache, a shape nobody built a core for
    (Si with the 4-body row: 10,10,10,10,5;1) -> nepmi_model_load compiles the core into the cache, says so on stderr, loads it;
    a second load finds the file"""
    import ctypes as C
    import shutil
    import gpumd_amd
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc on this host")
    assert not _core("10_10_10_10_5_1")
    monkeypatch.setenv("NEPMI_JIT", "1")
    monkeypatch.setenv("NEPMI_JIT_CACHE", str(tmp_path / "cache"))
    nep = H.golden("Si", "nep_4body.txt")
    m = gpumd_amd.Model(nep)
    err = capfd.readouterr().err
    assert "compiling the NEP kernels for this model's shape" in err and "no JIT core" not in err, err
    built = glob.glob(str(tmp_path / "cache" / "libnepmi_jit_10_10_10_10_5_1_*.so"))
    assert len(built) == 1, os.listdir(str(tmp_path / "cache"))
    assert not glob.glob(str(tmp_path / "cache" / "*.lock")) and not glob.glob(str(tmp_path / "cache" / "*.tmp*"))
    monkeypatch.setenv("NEPMI_JIT", "0")
    m0 = gpumd_amd.Model(nep)
    assert C.c_void_p.from_address(m.handle).value != C.c_void_p.from_address(m0.handle).value  # two libraries
    assert m.info.dim == m0.info.dim and m.info.num_L == 5
    m.close()
    m0.close()


def test_a_core_built_from_other_sources_is_refused(tmp_path):
    """A core is matched to the library by the hash of the sources BAKED INTO both at build time (capi_jit.h: nepmi_core_abi), not by
    what lies on disk at run time: the library's own hash is in the names of its prebuilt cores, and a file that carries the right
    name but was compiled from other text (here: a copy of a prebuilt core with the baked constant patched, under the name of a
    shape nobody built a core for) is not used -- its function table and handle layouts need not match."""
    import ctypes as C
    import struct
    import gpumd_amd
    cores = _core(CASES["Si-5body"])
    if not cores:
        pytest.skip("no prebuilt JIT core")

    class Abi(C.Structure):
        _fields_ = [("src_hash", C.c_uint64), ("api_bytes", C.c_uint64)]
    lib = C.CDLL(os.path.join(H.ROOT, "gpumd_amd", "lib", "libnepmi.so"))
    lib.nepmi_core_abi.restype = Abi
    mine = lib.nepmi_core_abi()
    assert mine.src_hash != 0 and ("%016x" % mine.src_hash) in os.path.basename(cores[0])
    blob = open(cores[0], "rb").read()
    key = struct.pack("<Q", mine.src_hash)
    assert blob.count(key) >= 1
    cache = tmp_path / "cache"
    cache.mkdir(mode=0o700)
    fake = cache / ("libnepmi_jit_10_10_10_10_5_1_%016x.so" % mine.src_hash)
    fake.write_bytes(blob.replace(key, struct.pack("<Q", mine.src_hash ^ 1)))
    assert not _core("10_10_10_10_5_1")
    # in a process of its own: a process keeps the cores it has loaded (test_first_load_of_a_new_shape_compiles_its_core builds the
    # genuine core of this shape)
    import subprocess
    import sys
    code = ("import ctypes as C, gpumd_amd\n"
            "m = gpumd_amd.Model(%r); p = gpumd_amd.Model(%r)\n"
            "print('same_library', C.c_void_p.from_address(m.handle).value == C.c_void_p.from_address(p.handle).value)\n"
            % (H.golden("Si", "nep_4body.txt"), H.golden("PbTe", "nep.txt")))
    env = dict(os.environ, NEPMI_JIT="2", NEPMI_JIT_CACHE=str(cache), PYTHONPATH=H.ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "built from other sources" in r.stderr and "zero-padded" in r.stderr, r.stderr
    assert "same_library True" in r.stdout, r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_force_parity_on_a_jit_core(name, monkeypatch):
    """the oracle comparison of tests/test_gpu_parity.py::test_force_parity, the engine served by the model's JIT core"""
    if not _core(CASES[name]):
        pytest.skip("no prebuilt JIT core for %s" % name)
    monkeypatch.setenv("NEPMI_JIT", "2")
    drv = H.GpuDriver()
    # (against the FP64 oracle: the suite's tolerance; against the FP32 oracle, two FP32 evaluations in different orders of sums
    # over up to 358 neighbours: twice the band)
    eng = P.check_force_parity(drv, name, f32_atol=4e-5)
    assert "shape=jit(" in eng.describe(), eng.describe()


@pytest.mark.gpu
def test_long_cutoff_model_in_the_scatter_form_on_its_jit_core(monkeypatch):
    """C_2024_NEP4 (rc 7 A: ~380 Verlet entries per atom, windows of 5,900 slots, bricks of three passes) on the one-lane window
    kernels with the force assembly as LDS scatter (147 KB of positions + accumulators per workgroup): the form bench.py's
    c2024 extras run, against the oracle"""
    if not _core(CASES["C-2024-window"]):
        pytest.skip("no prebuilt JIT core")
    monkeypatch.setenv("NEPMI_JIT", "2")
    drv = H.GpuDriver()
    eng = P.check_force_parity(drv, "C-2024-window", lanes=1, win_static=True, force_form=1)
    d = eng.describe()
    assert "shape=jit(" in d and "window=lds_static" in d and "lds_scatter_of_own_halves" in d, d


@pytest.mark.gpu
def test_run_loop_on_a_jit_core_matches_the_run_time_shape_kernels(monkeypatch):
    """20 NVE steps of diamond with the C-2024 model: JIT core vs the run-time-shape kernels of the library (same trajectory to
    FP32 rounding), and how much faster the step is"""
    name = "C-2024"
    if not _core(CASES[name]):
        pytest.skip("no prebuilt JIT core")
    import time
    nep = H.golden(*P.MODELS[name][0].split("/"))
    h, typ, x = H.diamond((16, 16, 16), 3.57, rattle=0.02, seed=3)
    n = len(typ)
    mass = np.full(n, H.MASS["C"])
    vel = H.maxwell_velocities(mass, 300.0, seed=5)
    out = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("NEPMI_JIT", mode)
        drv = H.GpuDriver()
        model = drv.model(nep)
        eng = drv.engine(model, n)
        d_t, d_m, d_x, d_v = drv.dev(typ), drv.dev(mass), drv.dev(x), drv.dev(vel)
        d_pe, d_f, d_w = drv.zeros(n), drv.zeros(3 * n), drv.zeros(9 * n)
        eng.force_compute(h, d_t, d_x, d_pe, d_f, d_w)
        eng.run_nve(h, d_t, d_m, 1.0 / H.TIME_UNIT, 5, d_x, d_v, d_pe, d_f, d_w)
        drv.sync()
        t0 = time.perf_counter()
        th = eng.run_nve(h, d_t, d_m, 1.0 / H.TIME_UNIT, 20, d_x, d_v, d_pe, d_f, d_w, thermo_every=20)
        drv.sync()
        out[mode] = (drv.host(d_x), np.asarray(th), (time.perf_counter() - t0) / 20 * 1e3, eng.describe())
    assert "shape=jit(" in out["2"][3] and "shape=generic" in out["0"][3]
    dx = np.abs(out["2"][0] - out["0"][0]).max()
    print("\n[JIT core] max |dx| after 25 steps vs the run-time-shape kernels: %.2e A" % dx)
    assert dx < 2e-5
    # (temperature, potential energy, P_xx: the core runs the window kernels, the run-time shape the gather kernels -- FP32 sums over
    # ~380 pairs per atom in different orders; the pressure is a difference of kinetic and virial parts)
    np.testing.assert_allclose(out["2"][1][:, :2], out["0"][1][:, :2], rtol=1e-5)
    np.testing.assert_allclose(out["2"][1][:, 2], out["0"][1][:, 2], rtol=5e-5)
    print("\n[JIT core] C-2024, %d atoms: %.3f ms/step on the JIT core, %.3f ms/step on the run-time-shape kernels (%.1fx)"
          % (n, out["2"][2], out["0"][2], out["0"][2] / out["2"][2]))
    assert out["2"][2] < out["0"][2]
