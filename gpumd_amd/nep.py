"""Host-side mirror of the reference's NEP potential object (src/force/nep.cuh `class NEP :
public Potential`) on top of the C ABI.  Arrays are torch CUDA tensors in GPUMD's SoA layout:

    position/velocity/force : float64 [3*N]  (x.. | y.. | z..)
    virial                  : float64 [9*N]  planes xx,yy,zz,xy,xz,yz,yx,zx,zy
    potential               : float64 [N];  type: int32 [N];  mass: float64 [N]
    box                     : 9 host floats ax,bx,cx,ay,by,cy,az,bz,cz (Box::cpu_h[0..8])
"""
import ctypes as C

import numpy as np

from . import _capi


def _h9(box):
    h = np.ascontiguousarray(np.asarray(box, dtype=np.float64).reshape(9))
    return h, h.ctypes.data_as(_capi.c_dp)


def _pbc3(pbc):
    p = np.ascontiguousarray(np.asarray(pbc, dtype=np.int32).reshape(3))
    return p, p.ctypes.data_as(_capi.c_ip)


class Model:
    """Parsed nep.txt (host only; mirrors the parsing half of NEP::NEP, nep.cu:100-377)."""

    def __init__(self, path, lib=None):
        from . import load_library
        self.lib = lib if lib is not None else load_library()
        self.path = path
        self.handle = self.lib.nepmi_model_load(path.encode())
        if not self.handle:
            raise _capi.NepmiError(-2, self.lib.nepmi_last_error().decode())
        self.info = _capi.NepmiInfo()
        _capi.check(self.lib, self.lib.nepmi_model_info(self.handle, C.byref(self.info)))
        self.symbols = [self.lib.nepmi_model_symbol(self.handle, t).decode() for t in range(self.info.num_types)]

    def close(self):
        if getattr(self, "handle", None):
            self.lib.nepmi_model_free(self.handle)
            self.handle = None

    def __del__(self):
        self.close()


class NEP:
    """One NEP potential instance bound to `n_atoms` atoms on the current HIP device.

    `compute(box, type, position, potential, force, virial)` has the argument meaning of
    Potential::compute (src/force/potential.cuh:37-43): it ADDS to the three output arrays.
    `force_compute` is Force::compute (force.cu:771-855): wrap, zero, compute.
    """

    def __init__(self, model, n_atoms, stream=None, pbc=(1, 1, 1), lib=None):
        import torch
        if isinstance(model, str):
            model = Model(model, lib=lib)
        self.model = model
        self.lib = model.lib
        if lib is None:
            if not torch.cuda.is_available():
                raise RuntimeError("gpumd_amd.NEP needs an MI355X (gfx950) device; there is no CPU fallback")
        self.n = int(n_atoms)
        self.pbc = tuple(int(p) for p in pbc)
        self._stream = stream
        sp = None
        if stream is not None:
            sp = C.c_void_p(stream.cuda_stream)
        elif lib is None:
            sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self.handle = self.lib.nepmi_engine_create(model.handle, self.n, sp)
        if not self.handle:
            raise _capi.NepmiError(-5, self.lib.nepmi_last_error().decode())

    # -- helpers -------------------------------------------------------------------------------
    @staticmethod
    def _ptr(t):
        if t is None:
            return None
        if hasattr(t, "data_ptr"):
            return C.c_void_p(t.data_ptr())
        return C.c_void_p(t.ctypes.data)  # numpy (emulator tests only)

    def _ck(self, st):
        return _capi.check(self.lib, st)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.nepmi_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    # -- the plugin surface --------------------------------------------------------------------
    def compute(self, box, type, position, potential, force, virial):
        _, hp = _h9(box)
        _, pp = _pbc3(self.pbc)
        self._ck(self.lib.nepmi_potential_compute(
            self.handle, hp, pp, self.n, self._ptr(type), self._ptr(position), self._ptr(potential),
            self._ptr(force), self._ptr(virial)))

    def compute_levels(self, box, pbc, n, type, position, level, potential, force, virial):
        """Potential::compute on a local (owned + ghost) system; see nepmi_potential_compute_levels."""
        _, hp = _h9(box)
        _, pp = _pbc3(pbc)
        self._ck(self.lib.nepmi_potential_compute_levels(
            self.handle, hp, pp, int(n), self._ptr(type), self._ptr(position), self._ptr(level),
            self._ptr(potential), self._ptr(force), self._ptr(virial)))

    def compute_levels_begin(self, box, pbc, n, type, position, level, potential, force, virial):
        """First half (owned atoms final, ghosts in flight) -> True if interior work was enqueued."""
        _, hp = _h9(box)
        _, pp = _pbc3(pbc)
        return self._ck(self.lib.nepmi_potential_compute_levels_begin(
            self.handle, hp, pp, int(n), self._ptr(type), self._ptr(position), self._ptr(level),
            self._ptr(potential), self._ptr(force), self._ptr(virial))) == 1

    def compute_levels_end(self, box, pbc, n, type, position, level, potential, force, virial):
        _, hp = _h9(box)
        _, pp = _pbc3(pbc)
        self._ck(self.lib.nepmi_potential_compute_levels_end(
            self.handle, hp, pp, int(n), self._ptr(type), self._ptr(position), self._ptr(level),
            self._ptr(potential), self._ptr(force), self._ptr(virial)))

    def set_external_skin(self, on=True):
        self._ck(self.lib.nepmi_engine_set_external_skin(self.handle, 1 if on else 0))

    def set_unwrapped(self, unwrapped=None):
        """Atom::unwrapped_position: a [3N] f64 array that every first half-step adds its drift to (None: off)."""
        self._unwrapped = unwrapped  # keep the storage alive while the engine points at it
        self._ck(self.lib.nepmi_engine_set_unwrapped(self.handle, self._ptr(unwrapped) if unwrapped is not None else None))

    def invalidate(self):
        self._ck(self.lib.nepmi_engine_invalidate(self.handle))

    def force_compute(self, box, type, position, potential, force, virial, n=None):
        """n: number of atoms in the arrays (default: the capacity the engine was created with)."""
        _, hp = _h9(box)
        _, pp = _pbc3(self.pbc)
        self._ck(self.lib.nepmi_force_compute(
            self.handle, hp, pp, self.n if n is None else int(n), self._ptr(type), self._ptr(position), self._ptr(potential),
            self._ptr(force), self._ptr(virial)))

    def apply_pbc(self, box, position):
        _, hp = _h9(box)
        _, pp = _pbc3(self.pbc)
        self._ck(self.lib.nepmi_apply_pbc(self.handle, hp, pp, self.n, self._ptr(position)))

    def zero_properties(self, potential, force, virial):
        """initialize_properties (force.cu:314-333)."""
        self._ck(self.lib.nepmi_zero_properties(self.handle, self.n, self._ptr(potential), self._ptr(force),
                                                self._ptr(virial)))

    def average_properties(self, denominator, potential, force, virial):
        """gpu_average_properties (force.cu:461-480): the closing division of the "average" multi-potential mode."""
        self._ck(self.lib.nepmi_average_properties(self.handle, self.n, float(denominator), self._ptr(potential),
                                                   self._ptr(force), self._ptr(virial)))

    def vv_step1(self, dt, mass, force, position, velocity):
        self._ck(self.lib.nepmi_vv_step1(self.handle, self.n, float(dt), self._ptr(mass), self._ptr(force),
                                         self._ptr(position), self._ptr(velocity)))

    def vv_step2(self, dt, mass, force, velocity):
        self._ck(self.lib.nepmi_vv_step2(self.handle, self.n, float(dt), self._ptr(mass), self._ptr(force),
                                         self._ptr(velocity)))

    def find_thermo(self, volume, mass, potential, velocity, virial, thermo8):
        self._ck(self.lib.nepmi_find_thermo(self.handle, self.n, float(volume), self._ptr(mass),
                                            self._ptr(potential), self._ptr(velocity), self._ptr(virial),
                                            self._ptr(thermo8)))

    def run_nve(self, box, type, mass, dt, nsteps, position, velocity, potential, force, virial,
                thermo_every=0):
        """Run::perform_a_run for `ensemble nve` -> thermo array [(nsteps // thermo_every), 8] (host)."""
        _, hp = _h9(box)
        _, pp = _pbc3(self.pbc)
        nrec = (nsteps // thermo_every) if thermo_every > 0 else 0
        th = np.zeros((max(nrec, 1), 8))
        self._ck(self.lib.nepmi_run_nve(
            self.handle, hp, pp, self.n, self._ptr(type), self._ptr(mass), float(dt), int(nsteps),
            self._ptr(position), self._ptr(velocity), self._ptr(potential), self._ptr(force),
            self._ptr(virial), int(thermo_every), th.ctypes.data_as(_capi.c_dp)))
        return th[:nrec]

    def run_nvt_ber(self, box, type, mass, dt, nsteps, t1, t2, t_coup, position, velocity, potential, force,
                    virial, thermo_every=0):
        """`ensemble nvt_ber T1 T2 T_coup` (Ensemble_BER) -> thermo array like run_nve."""
        _, hp = _h9(box)
        _, pp = _pbc3(self.pbc)
        nrec = (nsteps // thermo_every) if thermo_every > 0 else 0
        th = np.zeros((max(nrec, 1), 8))
        self._ck(self.lib.nepmi_run_nvt_ber(
            self.handle, hp, pp, self.n, self._ptr(type), self._ptr(mass), float(dt), int(nsteps), float(t1),
            float(t2), float(t_coup), self._ptr(position), self._ptr(velocity), self._ptr(potential),
            self._ptr(force), self._ptr(virial), int(thermo_every), th.ctypes.data_as(_capi.c_dp)))
        return th[:nrec]

    NHC_STATE_SIZE = 13

    def nhc_init(self, temperature, t_coup, dt, chain_state):
        """Ensemble_NHC constructor: chain_state = NHC_STATE_SIZE doubles of device memory."""
        self._ck(self.lib.nepmi_nhc_init(self.handle, self.n, float(temperature), float(t_coup), float(dt),
                                         self._ptr(chain_state)))

    def nhc_half_step(self, temperature, dt, thermo8, chain_state, velocity):
        self._ck(self.lib.nepmi_nhc_half_step(self.handle, self.n, float(temperature), float(dt),
                                              self._ptr(thermo8), self._ptr(chain_state), self._ptr(velocity)))

    def run_nvt_nhc(self, box, type, mass, dt, nsteps, t1, t2, t_coup, position, velocity, potential, force,
                    virial, thermo_every=0):
        """`ensemble nvt_nhc T1 T2 T_coup` (Ensemble_NHC) -> thermo array like run_nve."""
        _, hp = _h9(box)
        _, pp = _pbc3(self.pbc)
        nrec = (nsteps // thermo_every) if thermo_every > 0 else 0
        th = np.zeros((max(nrec, 1), 8))
        self._ck(self.lib.nepmi_run_nvt_nhc(
            self.handle, hp, pp, self.n, self._ptr(type), self._ptr(mass), float(dt), int(nsteps), float(t1),
            float(t2), float(t_coup), self._ptr(position), self._ptr(velocity), self._ptr(potential),
            self._ptr(force), self._ptr(virial), int(thermo_every), th.ctypes.data_as(_capi.c_dp)))
        return th[:nrec]

    def bdp_seed(self, seed):
        self._ck(self.lib.nepmi_bdp_seed(self.handle, int(seed)))

    def bdp_scale(self, temperature, t_coup, thermo8, velocity):
        self._ck(self.lib.nepmi_bdp_scale(self.handle, self.n, float(temperature), float(t_coup),
                                          self._ptr(thermo8), self._ptr(velocity)))

    def run_nvt_bdp(self, box, type, mass, dt, nsteps, t1, t2, t_coup, position, velocity, potential, force,
                    virial, thermo_every=0):
        """`ensemble nvt_bdp T1 T2 T_coup` (Ensemble_BDP) -> thermo array like run_nve."""
        _, hp = _h9(box)
        _, pp = _pbc3(self.pbc)
        nrec = (nsteps // thermo_every) if thermo_every > 0 else 0
        th = np.zeros((max(nrec, 1), 8))
        self._ck(self.lib.nepmi_run_nvt_bdp(
            self.handle, hp, pp, self.n, self._ptr(type), self._ptr(mass), float(dt), int(nsteps), float(t1),
            float(t2), float(t_coup), self._ptr(position), self._ptr(velocity), self._ptr(potential),
            self._ptr(force), self._ptr(virial), int(thermo_every), th.ctypes.data_as(_capi.c_dp)))
        return th[:nrec]

    def lan_seed(self, seed):
        """seed of the per-atom Langevin generators (the reference takes rand()); the next half-step initialises them"""
        self._ck(self.lib.nepmi_lan_seed(self.handle, int(seed)))

    def lan_half_step(self, temperature, t_coup, mass, velocity):
        """integrate_nvt_lan_half (Ensemble_LAN): Langevin kick of every atom + removal of the centre-of-mass velocity"""
        self._ck(self.lib.nepmi_lan_half_step(self.handle, self.n, float(temperature), float(t_coup), self._ptr(mass),
                                              self._ptr(velocity)))

    def run_nvt_lan(self, box, type, mass, dt, nsteps, t1, t2, t_coup, position, velocity, potential, force,
                    virial, thermo_every=0):
        """`ensemble nvt_lan T1 T2 T_coup` (Ensemble_LAN) -> thermo array like run_nve."""
        _, hp = _h9(box)
        _, pp = _pbc3(self.pbc)
        nrec = (nsteps // thermo_every) if thermo_every > 0 else 0
        th = np.zeros((max(nrec, 1), 8))
        self._ck(self.lib.nepmi_run_nvt_lan(
            self.handle, hp, pp, self.n, self._ptr(type), self._ptr(mass), float(dt), int(nsteps), float(t1),
            float(t2), float(t_coup), self._ptr(position), self._ptr(velocity), self._ptr(potential),
            self._ptr(force), self._ptr(virial), int(thermo_every), th.ctypes.data_as(_capi.c_dp)))
        return th[:nrec]

    def run_nvt_bao(self, box, type, mass, dt, nsteps, t1, t2, t_coup, position, velocity, potential, force,
                    virial, thermo_every=0):
        """`ensemble nvt_bao T1 T2 T_coup` (Ensemble_BAO, BAOAB Langevin) -> thermo array like run_nve."""
        _, hp = _h9(box)
        _, pp = _pbc3(self.pbc)
        nrec = (nsteps // thermo_every) if thermo_every > 0 else 0
        th = np.zeros((max(nrec, 1), 8))
        self._ck(self.lib.nepmi_run_nvt_bao(
            self.handle, hp, pp, self.n, self._ptr(type), self._ptr(mass), float(dt), int(nsteps), float(t1),
            float(t2), float(t_coup), self._ptr(position), self._ptr(velocity), self._ptr(potential),
            self._ptr(force), self._ptr(virial), int(thermo_every), th.ctypes.data_as(_capi.c_dp)))
        return th[:nrec]

    # -- diagnostics ---------------------------------------------------------------------------
    def neighbors(self, which, nn, nl, ld):
        return self._ck(self.lib.nepmi_neighbors_export(self.handle, int(which), self._ptr(nn), self._ptr(nl), int(ld)))

    def descriptors(self, q, fp):
        self._ck(self.lib.nepmi_descriptors_export(self.handle, self._ptr(q), self._ptr(fp)))

    def stats(self, with_lists=False):
        st = _capi.NepmiStats()
        self._ck(self.lib.nepmi_engine_stats(self.handle, 1 if with_lists else 0, C.byref(st)))
        return st

    def set_timing(self, on=True):
        """False/0 off, True/1 every kernel, 2 the force-assembly kernel only (see include/nepmi.h)."""
        self._ck(self.lib.nepmi_engine_set_timing(self.handle, int(on)))

    def set_option(self, name, value):
        """experiment / test switches by name (nepmi_engine_set_option; include/nepmi.h lists them)"""
        self._ck(self.lib.nepmi_engine_set_option(self.handle, name.encode(), float(value)))

    def set_tiles(self, mode=-1):
        """0/False: no LDS-window kernels; 1: radial pass only; 2/True: radial pass + force assembly;
        -1: chosen by the engine (times modes 2 and 1 once)."""
        mode = 2 if mode is True else 0 if mode is False else int(mode)
        self.set_option("tiles", mode)

    def set_win_lanes(self, lanes=0):
        """lanes per atom of the LDS-window kernels: 0 = by the number of bricks, or 1 / 2 / 4"""
        self.set_option("win_lanes", int(lanes))

    def set_force_form(self, mode=-1):
        """force assembly: -1 run loops scatter / per-call gather (default), 0 gather everywhere, 1 scatter wherever it applies
        (nepmi_engine_set_force_form)"""
        self._ck(self.lib.nepmi_engine_set_force_form(self.handle, int(mode)))

    def set_radial_mask(self, on=True):
        """scatter-form loop steps: inside bits over the packed Verlet words instead of a compacted radial list
        (option "radial_mask")"""
        self.set_option("radial_mask", 1 if on else 0)

    def set_radial_sync(self, on=True):
        """scatter-form loop steps: the radial list as wave-synchronous words (default) or the slot-major compact list
        (option "radial_sync")"""
        self.set_option("radial_sync", 1 if on else 0)

    def set_scatter_guard(self, ev_per_angstrom=64.0, hard_factor=0.0):
        """guard band of the scatter-form force assembly per pair half (test hook: options "scatter_guard", "scatter_guard_hard")"""
        self.set_option("scatter_guard", float(ev_per_angstrom))
        self.set_option("scatter_guard_hard", float(hard_factor))

    def set_virial_mode(self, mode=0):
        """per-call evaluations: 0 per-atom virials in the reference's attribution (default), 1 only the total has to be right
        (the scatter form of the force assembly where it applies): nepmi_engine_set_virial_mode"""
        self._ck(self.lib.nepmi_engine_set_virial_mode(self.handle, int(mode)))

    def set_brick_force(self, on=True):
        """fused angular kernel + scatter-form force assembly as ONE kernel per brick (default where it applies) or separately
        (option "brick_force")"""
        self.set_option("brick_force", 1 if on else 0)

    def set_angular_fused(self, on=True):
        """angular descriptor + ANN + partial angular forces in one kernel (default) or as separate kernels
        (option "angular_fused")"""
        self.set_option("angular_fused", 1 if on else 0)

    def describe(self):
        """the kernel forms the last force evaluation ran (counted rules of the engine, as text)"""
        import ctypes as C
        buf = C.create_string_buffer(512)
        n = self.lib.nepmi_engine_describe(self.handle, buf, 512)
        if n < 0:
            self._ck(n)
        return buf.value.decode()

    def set_stepwise_loops(self, on=True):
        """test hook: run_nvt_lan / run_nvt_bao as the stepwise sequence on the caller's arrays"""
        self.set_option("stepwise_loops", int(bool(on)))

    def set_win_static(self, on=True):
        """static window layout of the one-lane window kernels (default on); False = the scanned layout"""
        self.set_option("win_static", int(bool(on)))

    def set_mfma(self, on=True):
        """False / 0: per-atom ANN kernel; True / 1 (default): descriptor + ANN fused where the shape allows it, else the
        matrix-core ANN kernel; 2: the matrix-core kernel wherever it applies (no fusion)"""
        self.set_option("mfma", int(on))

    def set_temperature(self, temperature):
        """The `temperature` argument of NEP::compute(temperature, ...) for nep4[_zbl]_temperature models (no effect on
        plain models); stays in force for later compute / run calls."""
        self._ck(self.lib.nepmi_engine_set_temperature(self.handle, float(temperature)))

    def set_angular_recompute(self, mode=-1):
        self.set_option("angular_recompute", int(mode))

    def set_generic(self, on=True):
        self.set_option("generic", 1 if on else 0)
