"""Shared test utilities: ctypes bindings of the ORACLE libraries (test infrastructure only),
a minimal extended-XYZ reader, and synthetic-crystal builders.

Nothing here is imported by the product package ``gpumd_amd``.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


# input construction lives in the package (bench.py uses it without importing anything from tests/)
from gpumd_amd.structures import read_xyz_frames, replicate, soa, types_from_species  # noqa: E402,F401


# --------------------------------------------------------------------------------------------
# oracle: our plain-C restatement
# --------------------------------------------------------------------------------------------
class NepoInfo(C.Structure):
    _fields_ = [
        ("version", C.c_int), ("num_types", C.c_int), ("zbl_enabled", C.c_int),
        ("zbl_flexible", C.c_int), ("rc_radial_max", C.c_double), ("rc_angular_max", C.c_double),
        ("n_max_radial", C.c_int), ("n_max_angular", C.c_int), ("basis_size_radial", C.c_int),
        ("basis_size_angular", C.c_int), ("L_max", C.c_int), ("has_222", C.c_int),
        ("has_1111", C.c_int), ("num_L", C.c_int), ("dim", C.c_int), ("num_neurons", C.c_int),
        ("MN_radial", C.c_int), ("MN_angular", C.c_int), ("num_para", C.c_int)]


def build_oracle(ref=False):
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)
    if ref and os.path.isdir("/root/reference"):
        subprocess.run(["make", "-s", "-C", ORACLE_DIR, "_ref"], check=True)


_oracle_lib = None


def oracle_lib():
    global _oracle_lib
    if _oracle_lib is None:
        path = os.path.join(ORACLE_DIR, "libnep_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        L.nepo_model_load.restype = C.c_void_p
        L.nepo_model_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.nepo_model_free.argtypes = [C.c_void_p]
        L.nepo_model_info.argtypes = [C.c_void_p, C.POINTER(NepoInfo)]
        L.nepo_model_symbol.restype = C.c_char_p
        L.nepo_model_symbol.argtypes = [C.c_void_p, C.c_int]
        L.nepo_model_param.restype = C.c_double
        L.nepo_model_param.argtypes = [C.c_void_p, C.c_int]
        L.nepo_model_is_temperature.argtypes = [C.c_void_p]
        L.nepo_model_set_temperature.argtypes = [C.c_void_p, C.c_double]
        L.nepo_lists_build.restype = C.c_void_p
        L.nepo_lists_build.argtypes = [C.c_void_p, C.c_int, _ip, _dp, _ip, _dp, C.c_int]
        L.nepo_lists_path.argtypes = [C.c_void_p]
        L.nepo_lists_get.argtypes = [C.c_void_p, C.c_int, _ip, _ip, C.c_int]
        L.nepo_lists_free.argtypes = [C.c_void_p]
        L.nepo_compute.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _ip, _dp, _ip, _dp,
                                   _dp, _dp, _dp, _dp, _dp]
        L.nepo_apply_pbc.argtypes = [C.c_int, _dp, _ip, _dp]
        L.nepo_velocity_verlet.argtypes = [C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, _dp]
        L.nepo_thermo.argtypes = [C.c_int, C.c_double, _dp, _dp, _dp, _dp, _dp]
        L.nepo_nhc_init.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, _dp]
        L.nepo_nhc.argtypes = [_dp, C.c_double, C.c_double, C.c_double, C.c_double]
        L.nepo_nhc.restype = C.c_double
        L.nepo_bdp_sizeof.restype = C.c_int
        L.nepo_bdp_seed.argtypes = [C.c_void_p, C.c_uint]
        L.nepo_bdp_factor.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
        L.nepo_bdp_factor.restype = C.c_double
        L.nepo_run_nve.argtypes = [C.c_void_p, C.c_int, C.c_int, _ip, _dp, _ip, _dp, C.c_double,
                                   C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
        _oracle_lib = L
    return _oracle_lib


class Oracle:
    """Our CPU restatement of the reference NEP path (oracle/nep_oracle.c)."""

    def __init__(self, nep_txt):
        L = oracle_lib()
        err = C.create_string_buffer(512)
        self.h = L.nepo_model_load(nep_txt.encode(), err, 512)
        if not self.h:
            raise RuntimeError("oracle: " + err.value.decode())
        self.info = NepoInfo()
        L.nepo_model_info(self.h, C.byref(self.info))
        self.symbols = [L.nepo_model_symbol(self.h, t).decode() for t in range(self.info.num_types)]

    def __del__(self):
        if getattr(self, "h", None):
            oracle_lib().nepo_model_free(self.h)
            self.h = None

    def set_temperature(self, temperature):
        """the temperature argument of NEP::compute(temperature, ...) (temperature-dependent models)"""
        oracle_lib().nepo_model_set_temperature(self.h, float(temperature))

    def compute(self, typ, h, pos_soa, pbc=(1, 1, 1), precision=32, path=-1, stages=False):
        L = oracle_lib()
        n = len(typ)
        typ = np.ascontiguousarray(typ, dtype=np.int32)
        h = np.ascontiguousarray(h, dtype=np.float64)
        pbc = np.ascontiguousarray(pbc, dtype=np.int32)
        pos = np.ascontiguousarray(pos_soa, dtype=np.float64)
        pe = np.zeros(n)
        f = np.zeros(3 * n)
        v = np.zeros(9 * n)
        q = np.zeros(self.info.dim * n) if stages else None
        fp = np.zeros(self.info.dim * n) if stages else None
        st = L.nepo_compute(self.h, precision, path, n, _p(typ, _ip), _p(h, _dp), _p(pbc, _ip),
                            _p(pos, _dp), _p(pe, _dp), _p(f, _dp), _p(v, _dp), _p(q, _dp), _p(fp, _dp))
        if st != 0:
            raise RuntimeError("oracle compute failed: %d" % st)
        if stages:
            return pe, f, v, q.reshape(self.info.dim, n), fp.reshape(self.info.dim, n)
        return pe, f, v

    def lists(self, typ, h, pos_soa, pbc=(1, 1, 1), path=-1):
        """-> dict which -> (nn[N], nl[ld,N]) ; which in skin/radial/angular"""
        L = oracle_lib()
        n = len(typ)
        typ = np.ascontiguousarray(typ, dtype=np.int32)
        h = np.ascontiguousarray(h, dtype=np.float64)
        pbc = np.ascontiguousarray(pbc, dtype=np.int32)
        pos = np.ascontiguousarray(pos_soa, dtype=np.float64)
        hl = L.nepo_lists_build(self.h, n, _p(typ, _ip), _p(h, _dp), _p(pbc, _ip), _p(pos, _dp), path)
        if not hl:
            raise RuntimeError("oracle list build failed")
        out = {"path": L.nepo_lists_path(hl)}
        for which, name in ((0, "skin"), (1, "radial"), (2, "angular")):
            nn = np.zeros(n, dtype=np.int32)
            mx = L.nepo_lists_get(hl, which, _p(nn, _ip), None, 0)
            if mx < 0:
                continue
            nl = np.full((max(mx, 1), n), -1, dtype=np.int32)
            L.nepo_lists_get(hl, which, _p(nn, _ip), _p(nl, _ip), max(mx, 1))
            out[name] = (nn, nl)
        L.nepo_lists_free(hl)
        return out

    def run_nve(self, typ, h, pos_soa, vel_soa, mass, dt, nsteps, pbc=(1, 1, 1), precision=32):
        L = oracle_lib()
        n = len(typ)
        typ = np.ascontiguousarray(typ, dtype=np.int32)
        h = np.ascontiguousarray(h, dtype=np.float64)
        pbc = np.ascontiguousarray(pbc, dtype=np.int32)
        pos = np.array(pos_soa, dtype=np.float64)
        vel = np.array(vel_soa, dtype=np.float64)
        mass = np.ascontiguousarray(mass, dtype=np.float64)
        pe = np.zeros(n)
        f = np.zeros(3 * n)
        v = np.zeros(9 * n)
        th = np.zeros(8 * nsteps)
        rb = L.nepo_run_nve(self.h, precision, n, _p(typ, _ip), _p(h, _dp), _p(pbc, _ip), _p(mass, _dp),
                            dt, nsteps, _p(pos, _dp), _p(vel, _dp), _p(pe, _dp), _p(f, _dp), _p(v, _dp),
                            _p(th, _dp))
        if rb < 0:
            raise RuntimeError("oracle run failed: %d" % rb)
        return dict(pos=pos, vel=vel, pe=pe, force=f, virial=v, thermo=th.reshape(nsteps, 8), rebuilds=rb)


def oracle_apply_pbc(h, pos_soa, pbc=(1, 1, 1)):
    L = oracle_lib()
    pos = np.array(pos_soa, dtype=np.float64)
    n = pos.size // 3
    h = np.ascontiguousarray(h, dtype=np.float64)
    pbc = np.ascontiguousarray(pbc, dtype=np.int32)
    L.nepo_apply_pbc(n, _p(h, _dp), _p(pbc, _ip), _p(pos, _dp))
    return pos


def oracle_thermo(volume, mass, pe, vel, virial):
    L = oracle_lib()
    n = len(mass)
    th = np.zeros(8)
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (mass, pe, vel, virial)]
    L.nepo_thermo(n, volume, _p(a[0], _dp), _p(a[1], _dp), _p(a[2], _dp), _p(a[3], _dp), _p(th, _dp))
    return th


# --------------------------------------------------------------------------------------------
# oracle/_ref: the reference's own NEP_CPU compiled in place
# --------------------------------------------------------------------------------------------
_ref_lib = None


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libnepcpu_ref.so"))


def ref_lib():
    global _ref_lib
    if _ref_lib is None:
        L = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libnepcpu_ref.so"))
        L.nepref_create.restype = C.c_void_p
        L.nepref_create.argtypes = [C.c_char_p]
        L.nepref_destroy.argtypes = [C.c_void_p]
        L.nepref_compute.argtypes = [C.c_void_p, C.c_int, _ip, _dp, _dp, _dp, _dp, _dp]
        L.nepref_descriptor.argtypes = [C.c_void_p, C.c_int, _ip, _dp, _dp, _dp]
        L.nepref_neighbors.argtypes = [C.c_void_p, C.c_int, C.c_int, _ip, _ip, C.c_int]
        L.nepref_info.argtypes = [C.c_void_p, _dp, _dp, _ip, _ip]
        _ref_lib = L
    return _ref_lib


class RefNepCpu:
    def __init__(self, nep_txt):
        self.L = ref_lib()
        self.h = self.L.nepref_create(nep_txt.encode())
        rr, ra = C.c_double(), C.c_double()
        dim, nt = C.c_int(), C.c_int()
        self.L.nepref_info(self.h, C.byref(rr), C.byref(ra), C.byref(dim), C.byref(nt))
        self.dim = dim.value

    def __del__(self):
        if getattr(self, "h", None):
            self.L.nepref_destroy(self.h)
            self.h = None

    def compute(self, typ, h, pos_soa):
        n = len(typ)
        typ = np.ascontiguousarray(typ, dtype=np.int32)
        h = np.ascontiguousarray(h, dtype=np.float64)
        pos = np.ascontiguousarray(pos_soa, dtype=np.float64)
        pe = np.zeros(n)
        f = np.zeros(3 * n)
        v = np.zeros(9 * n)
        self.L.nepref_compute(self.h, n, _p(typ, _ip), _p(h, _dp), _p(pos, _dp), _p(pe, _dp), _p(f, _dp), _p(v, _dp))
        return pe, f, v

    def descriptor(self, typ, h, pos_soa):
        n = len(typ)
        typ = np.ascontiguousarray(typ, dtype=np.int32)
        h = np.ascontiguousarray(h, dtype=np.float64)
        pos = np.ascontiguousarray(pos_soa, dtype=np.float64)
        d = np.zeros(self.dim * n)
        self.L.nepref_descriptor(self.h, n, _p(typ, _ip), _p(h, _dp), _p(pos, _dp), _p(d, _dp))
        return d.reshape(self.dim, n)

    def neighbors(self, n, which, ld=512):
        nn = np.zeros(n, dtype=np.int32)
        nl = np.full((ld, n), -1, dtype=np.int32)
        self.L.nepref_neighbors(self.h, which, n, _p(nn, _ip), _p(nl, _ip), ld)
        return nn, nl


def neighbor_sets(nn, nl):
    """per-atom sorted tuple of neighbour indices (multiset, since small boxes repeat images)"""
    return [tuple(sorted(nl[:nn[i], i].tolist())) for i in range(len(nn))]


# --------------------------------------------------------------------------------------------
# synthetic configurations (seeded) used by the emulator and the GPU parity tests alike
# --------------------------------------------------------------------------------------------
def golden(*parts):
    return os.path.join(GOLDEN, *parts)


def pbte_supercell(reps, rattle=0.03, seed=1, symbols=("Te", "Pb"), num_types=None):
    """replicate (replicate.cu:50-71 order) of the 250-atom triclinic PbTe cell + Gaussian rattle.
    For models with other species the types are assigned round-robin over `num_types`."""
    fr = read_xyz_frames(golden("PbTe", "model.xyz"))[0]
    if num_types is None:
        typ0 = types_from_species(fr["species"], list(symbols))
    else:
        typ0 = (np.arange(fr["n"]) * 7 % num_types).astype(np.int32)
    h, typ, pos = replicate(fr["h"], typ0, fr["pos"], reps)
    rng = np.random.default_rng(seed)
    pos = pos + rng.normal(0.0, rattle, pos.shape)
    x = oracle_apply_pbc(h, soa(pos))
    return h, typ.astype(np.int32), x


def rocksalt_orthogonal(cells, a=6.5704, rattle=0.02, seed=3):
    """orthogonal rock-salt PbTe (SURVEY 8d.3 variant): 8 atoms per conventional cell."""
    basis = np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5],
                      [.5, 0, 0], [0, .5, 0], [0, 0, .5], [.5, .5, .5]]) * a
    typ0 = np.array([1, 1, 1, 1, 0, 0, 0, 0], dtype=np.int32)  # Pb = 1, Te = 0 (nep4 2 Te Pb)
    h0 = np.diag([a, a, a]).reshape(9)
    h, typ, pos = replicate(h0, typ0, basis, cells)
    rng = np.random.default_rng(seed)
    pos = pos + rng.normal(0.0, rattle, pos.shape)
    x = oracle_apply_pbc(h, soa(pos))
    return h, typ.astype(np.int32), x


def fcc_alloy(cells, a, num_types, rattle=0.05, seed=5):
    basis = np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]]) * a
    h0 = np.diag([a, a, a]).reshape(9)
    h, _, pos = replicate(h0, np.zeros(4, dtype=np.int32), basis, cells)
    rng = np.random.default_rng(seed)
    typ = rng.integers(0, num_types, len(pos)).astype(np.int32)
    pos = pos + rng.normal(0.0, rattle, pos.shape)
    x = oracle_apply_pbc(h, soa(pos))
    return h, typ, x


def diamond(cells, a, rattle=0.03, seed=9):
    fcc = np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]])
    basis = np.concatenate([fcc, fcc + 0.25]) * a
    h0 = np.diag([a, a, a]).reshape(9)
    h, typ, pos = replicate(h0, np.zeros(8, dtype=np.int32), basis, cells)
    rng = np.random.default_rng(seed)
    pos = pos + rng.normal(0.0, rattle, pos.shape)
    x = oracle_apply_pbc(h, soa(pos))
    return h, typ.astype(np.int32), x


from gpumd_amd.structures import K_B, MASS, TIME_UNIT, maxwell_velocities  # noqa: E402,F401


class EmuDriver:
    name = "emu"

    def __init__(self):
        from gpumd_amd import _capi
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")], check=True)
        self.lib = _capi.bind(C.CDLL(os.path.join(ROOT, "tests", "emu", "libnepmi_emu.so")))

    def model(self, nep_txt):
        from gpumd_amd.nep import Model
        return Model(nep_txt, lib=self.lib)

    def engine(self, model, n, pbc=(1, 1, 1)):
        from gpumd_amd.nep import NEP
        return NEP(model, n, pbc=pbc, lib=self.lib)

    def dev(self, a):
        return np.array(a, copy=True)

    def zeros(self, n, dtype=np.float64):
        return np.zeros(n, dtype=dtype)

    def host(self, a):
        return np.array(a, copy=True)

    def sync(self):
        pass


class GpuDriver:
    name = "gpu"

    def __init__(self):
        import torch
        import gpumd_amd
        assert torch.cuda.is_available()
        self.torch = torch
        self.devi = torch.device("cuda:0")
        self.lib = gpumd_amd.load_library()

    def model(self, nep_txt):
        import gpumd_amd
        return gpumd_amd.Model(nep_txt)

    def engine(self, model, n, pbc=(1, 1, 1)):
        import gpumd_amd
        return gpumd_amd.NEP(model, n, pbc=pbc)

    def dev(self, a):
        t = self.torch.from_numpy(np.ascontiguousarray(a)).to(self.devi)
        self.torch.cuda.current_stream().synchronize()
        return t

    def zeros(self, n, dtype=np.float64):
        tdt = {np.float64: self.torch.float64, np.float32: self.torch.float32, np.int32: self.torch.int32}[dtype]
        t = self.torch.zeros(n, dtype=tdt, device=self.devi)
        # the fill runs on torch's current stream, the engines of the in-process multi-rank tests on streams of their own (not
        # ordered against it): without this a gather into the fresh array could be overtaken by its own zero fill
        self.torch.cuda.current_stream().synchronize()
        return t

    def host(self, a):
        self.torch.cuda.synchronize()
        return a.cpu().numpy()

    def sync(self):
        self.torch.cuda.synchronize()


def engine_force(drv, eng, h, typ, x):
    """Force::compute through the C ABI -> (wrapped pos, pe, f, v) as numpy."""
    n = len(typ)
    d_t, d_x = drv.dev(typ), drv.dev(x)
    d_pe, d_f, d_v = drv.zeros(n), drv.zeros(3 * n), drv.zeros(9 * n)
    eng.force_compute(h, d_t, d_x, d_pe, d_f, d_v, n=n)
    return drv.host(d_x), drv.host(d_pe), drv.host(d_f), drv.host(d_v)


def engine_lists(drv, eng, n, which, ld=512):
    nn = drv.zeros(n, dtype=np.int32)
    nl = drv.zeros(ld * n, dtype=np.int32)
    mx = eng.neighbors(which, nn, nl, ld)
    return mx, drv.host(nn), drv.host(nl).reshape(ld, n)


def assert_lists_equal(nn, nl, onn, onl):
    assert np.array_equal(nn, onn)
    ld = min(nl.shape[0], onl.shape[0])
    assert onn.max() <= ld
    mask = np.arange(ld)[:, None] < nn[None, :]
    assert np.array_equal(np.where(mask, nl[:ld], -1), np.where(mask, onl[:ld], -1))


# the reference's own regression tolerances, tests_pytest/conftest.py:51-60
TOL = dict(energy_rtol=1e-5, energy_atol=1e-8, force_rtol=1e-4, force_atol=1e-6, virial_rtol=1e-4, virial_atol=1e-6)


# --------------------------------------------------------------------------------------------
# Tersoff-1989 oracle (oracle/tersoff_oracle.c)
# --------------------------------------------------------------------------------------------
_terso_lib = None


def terso_lib():
    global _terso_lib
    if _terso_lib is None:
        path = os.path.join(ORACLE_DIR, "libtersoff_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        L.terso_load.restype = C.c_void_p
        L.terso_load.argtypes = [C.c_char_p]
        L.terso_free.argtypes = [C.c_void_p]
        L.terso_rc.restype = C.c_double
        L.terso_rc.argtypes = [C.c_void_p]
        L.terso_compute.argtypes = [C.c_void_p, C.c_int, _ip, _dp, _ip, _dp, _dp, _dp, _dp, _ip, _ip, C.c_int]
        L.terso_params.argtypes = [C.c_void_p, _dp]
        _terso_lib = L
    return _terso_lib


class TersoffOracle:
    def __init__(self, path):
        self.L = terso_lib()
        self.h = self.L.terso_load(path.encode())
        if not self.h:
            raise RuntimeError("tersoff oracle: cannot load " + path)
        self.rc = self.L.terso_rc(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.terso_free(self.h)
            self.h = None

    def compute(self, typ, h, pos_soa, pbc=(1, 1, 1), lists=False):
        n = len(typ)
        typ = np.ascontiguousarray(typ, dtype=np.int32)
        h = np.ascontiguousarray(h, dtype=np.float64)
        pbc = np.ascontiguousarray(pbc, dtype=np.int32)
        pos = np.ascontiguousarray(pos_soa, dtype=np.float64)
        pe, f, v = np.zeros(n), np.zeros(3 * n), np.zeros(9 * n)
        nn = np.zeros(n, dtype=np.int32)
        nl = np.full((64, n), -1, dtype=np.int32)
        mx = self.L.terso_compute(self.h, n, _p(typ, _ip), _p(h, _dp), _p(pbc, _ip), _p(pos, _dp), _p(pe, _dp),
                                  _p(f, _dp), _p(v, _dp), _p(nn, _ip), _p(nl, _ip), 64)
        if mx < 0:
            raise RuntimeError("tersoff oracle failed")
        if lists:
            return pe, f, v, nn, nl[:max(mx, 1)]
        return pe, f, v
