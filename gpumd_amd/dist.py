"""Python mirror of the multi-GPU entry points of libnepmi.so (include/nepmi.h, nepmi_dist_*): one process per GPU,
spatial domain decomposition with ghost exchange in the C++ driver (gpumd_amd/csrc/dist_impl.h).  This module only
passes pointers: nothing of the per-step work happens in Python.

    tr = Transport.rccl(lib, rank, world, broadcast=...)     # device buffers over xGMI
    tr = Transport.tcp(lib, "127.0.0.1", port, rank, world)  # host sockets (tests, ranks sharing a GPU)
    md = DistMD(model, tr, h9, pbc, grid)
    md.setup(type, mass, pos, vel)       # this rank's share of the atoms (device arrays), any positions
    md.run("nve", dt, nsteps)            # or nvt_ber / nvt_nhc / nvt_bdp with t1, t2, t_coup
"""
import ctypes as C

import numpy as np

from . import _capi

ENSEMBLES = {"nve": 0, "nvt_ber": 1, "nvt_nhc": 2, "nvt_bdp": 3, "nvt_lan": 4, "nvt_bao": 5}


def choose_grid(world):
    """Process grid for `world` ranks: as cubic as possible (all 7 peers of a 2x2x2 grid are direct xGMI links
    on an 8-GPU node)."""
    best = (world, 1, 1)
    for a in range(1, world + 1):
        if world % a:
            continue
        for b in range(1, world // a + 1):
            if (world // a) % b:
                continue
            c = world // a // b
            g = tuple(sorted((a, b, c), reverse=True))
            if max(g) - min(g) < max(best) - min(best):
                best = g
    return best


class Transport:
    def __init__(self, lib, struct):
        self.lib = lib
        self.struct = struct

    @classmethod
    def tcp(cls, lib, master_addr, port, rank, nranks):
        t = _capi.NepmiTransport()
        _capi.check(lib, lib.nepmi_transport_tcp(master_addr.encode(), int(port), int(rank), int(nranks), C.byref(t)))
        return cls(lib, t)

    @classmethod
    def rccl(cls, lib, rank, nranks, broadcast):
        """broadcast(bytes_or_None) -> bytes: hands rank 0's 128-byte id to every rank (e.g. over torch.distributed)."""
        buf = C.create_string_buffer(128)
        if rank == 0:
            _capi.check(lib, lib.nepmi_transport_rccl_id(buf))
        ident = broadcast(buf.raw if rank == 0 else None)
        t = _capi.NepmiTransport()
        _capi.check(lib, lib.nepmi_transport_rccl(ident, int(rank), int(nranks), C.byref(t)))
        return cls(lib, t)

    def rccl_stats(self, time_every=0, reset=False):
        """counters of an RCCL transport (nepmi_transport_rccl_stats) as a dict, or None for any other transport; time_every > 0:
        from now on every time_every-th grouped exchange is bracketed by HIP events on its stream"""
        out = _capi.NepmiRcclStats()
        if self.lib.nepmi_transport_rccl_stats(C.byref(self.struct), int(time_every), 1 if reset else 0, C.byref(out)) != 0:
            return None
        return {k: getattr(out, k) for k, _ in out._fields_}

    def close(self):
        if self.struct is not None:
            self.lib.nepmi_transport_destroy(C.byref(self.struct))
            self.struct = None


class DistMD:
    def __init__(self, model, transport, h9, pbc, grid, stream=None, ghost_mode=None):
        """ghost_mode: None / -1 the counted rule, 0 forward ghosts (shell 2 (rc + skin), one exchange per step),
        1 reverse ghosts (shell rc + skin, forces of the ghosts travel back): nepmi_dist_set_ghost_mode"""
        self.lib = model.lib
        self.model = model
        self.transport = transport
        h = np.ascontiguousarray(np.asarray(h9, dtype=np.float64).reshape(9))
        p = np.ascontiguousarray(np.asarray(pbc, dtype=np.int32).reshape(3))
        g = np.ascontiguousarray(np.asarray(grid, dtype=np.int32).reshape(3))
        sp = C.c_void_p(stream.cuda_stream) if stream is not None else None
        self.handle = self.lib.nepmi_dist_create(
            model.handle, C.byref(transport.struct), h.ctypes.data_as(C.POINTER(C.c_double)),
            p.ctypes.data_as(C.POINTER(C.c_int)), g.ctypes.data_as(C.POINTER(C.c_int)), sp)
        if not self.handle:
            raise _capi.NepmiError(-5, self.lib.nepmi_last_error().decode())
        if ghost_mode is not None:
            self._ck(self.lib.nepmi_dist_set_ghost_mode(self.handle, int(ghost_mode)))

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
            self.lib.nepmi_dist_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    def setup(self, type, mass, pos, vel, ids=None):
        self._ck(self.lib.nepmi_dist_setup(self.handle, len(type), self._ptr(type), self._ptr(mass), self._ptr(pos),
                                           self._ptr(vel), self._ptr(ids)))

    def compute(self):
        self._ck(self.lib.nepmi_dist_compute(self.handle))

    def run(self, ensemble, dt, nsteps, t1=0.0, t2=0.0, t_coup=1.0, thermo_every=0):
        nrec = nsteps // thermo_every if thermo_every > 0 else 0
        th = np.zeros((max(nrec, 1), 8), dtype=np.float64)
        self._ck(self.lib.nepmi_dist_run(self.handle, ENSEMBLES[ensemble], float(dt), int(nsteps), float(t1), float(t2),
                                         float(t_coup), int(thermo_every), th.ctypes.data_as(C.POINTER(C.c_double))))
        return th[:nrec]

    def thermo(self):
        th = np.zeros(8, dtype=np.float64)
        self._ck(self.lib.nepmi_dist_thermo(self.handle, th.ctypes.data_as(C.POINTER(C.c_double))))
        return th

    def bdp_seed(self, seed):
        self._ck(self.lib.nepmi_dist_bdp_seed(self.handle, int(seed)))

    def lan_seed(self, seed):
        """seed of the per-atom Langevin generators (ensemble nvt_lan); the same value on every rank"""
        self._ck(self.lib.nepmi_dist_lan_seed(self.handle, int(seed)))

    def set_overlap(self, on=True):
        self._ck(self.lib.nepmi_dist_set_overlap(self.handle, 1 if on else 0))

    def num_overlapped_reverse(self):
        """steps whose interior force assembly ran while the ghosts' partial forces travelled (reverse ghosts + overlap)"""
        return int(self.lib.nepmi_dist_num_overlapped_reverse(self.handle))

    def info(self):
        assert self.lib.nepmi_dist_info_bytes() == C.sizeof(_capi.NepmiDistInfo)
        out = _capi.NepmiDistInfo()
        self._ck(self.lib.nepmi_dist_get_info(self.handle, C.byref(out)))
        return out

    def gather_owned(self, ids, pos, vel, force, pe=None, virial=None):
        self._ck(self.lib.nepmi_dist_gather_owned(self.handle, self._ptr(ids), self._ptr(pos), self._ptr(vel),
                                                  self._ptr(force), self._ptr(pe), self._ptr(virial)))

    def gather_global(self, root, pos=None, vel=None, force=None, pe=None, virial=None):
        """every atom of the system on rank `root`, ordered by global id (device arrays with n_total entries per plane on the
        root, ignored elsewhere); collective (nepmi_dist_gather_global)"""
        self._ck(self.lib.nepmi_dist_gather_global(self.handle, int(root), self._ptr(pos), self._ptr(vel), self._ptr(force),
                                                   self._ptr(pe), self._ptr(virial)))

    def engine_stats(self, with_lists=False):
        e = self.lib.nepmi_dist_engine(self.handle)
        st = _capi.NepmiStats()
        self._ck(self.lib.nepmi_engine_stats(e, 1 if with_lists else 0, C.byref(st)))
        return st

    def engine_set_timing(self, mode):
        self._ck(self.lib.nepmi_engine_set_timing(self.lib.nepmi_dist_engine(self.handle), int(mode)))

    def engine_describe(self):
        """the kernel forms the local engine's last force evaluation ran (nepmi_engine_describe)"""
        buf = C.create_string_buffer(512)
        n = self.lib.nepmi_engine_describe(self.lib.nepmi_dist_engine(self.handle), buf, 512)
        return buf.value.decode() if n >= 0 else ""
