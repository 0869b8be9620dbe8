"""ctypes declarations of the C ABI in include/nepmi.h (one place, so signatures cannot drift).

`bind(cdll)` only attaches argtypes/restype; it does not decide which library is loaded.  The
product loads gpumd_amd/lib/libnepmi.so (gfx950 code objects) through `gpumd_amd.load_library()`.
"""
import ctypes as C

c_i64 = C.c_int64
c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
VP = C.c_void_p  # device pointers travel as integers


class NepmiInfo(C.Structure):
    _fields_ = [
        ("version", C.c_int), ("num_types", C.c_int), ("zbl_enabled", C.c_int), ("zbl_flexible", C.c_int),
        ("zbl_rc_inner", C.c_double), ("zbl_rc_outer", C.c_double),
        ("rc_radial", C.c_double), ("rc_angular", C.c_double),
        ("MN_radial", C.c_int), ("MN_angular", C.c_int),
        ("n_max_radial", C.c_int), ("n_max_angular", C.c_int),
        ("basis_size_radial", C.c_int), ("basis_size_angular", C.c_int),
        ("L_max", C.c_int), ("has_q_222", C.c_int), ("has_q_1111", C.c_int), ("num_L", C.c_int),
        ("dim", C.c_int), ("num_neurons", C.c_int), ("num_para", C.c_int),
        ("has_q_112", C.c_int), ("has_q_123", C.c_int), ("has_q_233", C.c_int), ("has_q_134", C.c_int),
        ("model_type", C.c_int)]


class NepmiStats(C.Structure):
    _fields_ = [
        ("num_compute", c_i64), ("num_rebuild", c_i64),
        ("max_nn_skin", C.c_int), ("max_nn_radial", C.c_int), ("max_nn_angular", C.c_int),
        ("mean_nn_radial", C.c_double), ("mean_nn_angular", C.c_double),
        ("ms_force_last", C.c_double), ("ms_kernel", C.c_double * 8),
        ("ms_kernel_sum", C.c_double * 8), ("launches", c_i64 * 8), ("radial_tiles", C.c_int),
        ("discarded_steps", c_i64)]


class NepmiMsg(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("bytes", c_i64), ("peer", C.c_int)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(NepmiMsg), C.c_int, C.POINTER(NepmiMsg), C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, c_i64, C.c_int, C.c_int, C.c_void_p)
DESTROY_FN = C.CFUNCTYPE(None, C.c_void_p)


class NepmiTransport(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int), ("nranks", C.c_int), ("device_buffers", C.c_int),
                ("exchange", EXCHANGE_FN), ("allreduce", ALLREDUCE_FN), ("destroy", DESTROY_FN)]


class NepmiDistInfo(C.Structure):
    _fields_ = [("n_owned", c_i64), ("n_local", c_i64), ("n_total", c_i64), ("num_decompositions", c_i64),
                ("num_steps", c_i64), ("num_overlapped", c_i64), ("decompose_ms", C.c_double),
                ("reverse_ghosts", c_i64), ("num_range_handovers", c_i64)]


class NepmiRcclStats(C.Structure):
    _fields_ = [("comm_nranks", c_i64), ("comm_rank", c_i64), ("exchanges", c_i64), ("messages", c_i64),
                ("bytes_sent", c_i64), ("bytes_received", c_i64), ("allreduces", c_i64), ("timed_exchanges", c_i64),
                ("us_per_timed_exchange", C.c_double)]


# every symbol include/nepmi.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "nepmi_last_error": (C.c_char_p, []),
    "nepmi_version": (C.c_int, []),
    "nepmi_model_load": (VP, [C.c_char_p]),
    "nepmi_model_free": (None, [VP]),
    "nepmi_model_info": (C.c_int, [VP, C.POINTER(NepmiInfo)]),
    "nepmi_model_symbol": (C.c_char_p, [VP, C.c_int]),
    "nepmi_engine_create": (VP, [VP, c_i64, VP]),
    "nepmi_engine_destroy": (None, [VP]),
    "nepmi_force_compute": (C.c_int, [VP, c_dp, c_ip, c_i64, VP, VP, VP, VP, VP]),
    "nepmi_potential_compute": (C.c_int, [VP, c_dp, c_ip, c_i64, VP, VP, VP, VP, VP]),
    "nepmi_potential_compute_levels": (C.c_int, [VP, c_dp, c_ip, c_i64, VP, VP, VP, VP, VP, VP]),
    "nepmi_potential_compute_levels_begin": (C.c_int, [VP, c_dp, c_ip, c_i64, VP, VP, VP, VP, VP, VP]),
    "nepmi_potential_compute_levels_end": (C.c_int, [VP, c_dp, c_ip, c_i64, VP, VP, VP, VP, VP, VP]),
    "nepmi_engine_invalidate": (C.c_int, [VP]),
    "nepmi_engine_set_external_skin": (C.c_int, [VP, C.c_int]),
    "nepmi_engine_set_unwrapped": (C.c_int, [VP, VP]),
    "nepmi_apply_pbc": (C.c_int, [VP, c_dp, c_ip, c_i64, VP]),
    "nepmi_zero_properties": (C.c_int, [VP, c_i64, VP, VP, VP]),
    "nepmi_average_properties": (C.c_int, [VP, c_i64, C.c_double, VP, VP, VP]),
    "nepmi_vv_step1": (C.c_int, [VP, c_i64, C.c_double, VP, VP, VP, VP]),
    "nepmi_vv_step2": (C.c_int, [VP, c_i64, C.c_double, VP, VP, VP]),
    "nepmi_find_thermo": (C.c_int, [VP, c_i64, C.c_double, VP, VP, VP, VP, VP]),
    "nepmi_run_nve": (C.c_int, [VP, c_dp, c_ip, c_i64, VP, VP, C.c_double, c_i64, VP, VP, VP, VP, VP,
                                c_i64, c_dp]),
    "nepmi_berendsen_scale": (C.c_int, [VP, c_i64, C.c_double, C.c_double, VP, VP]),
    "nepmi_run_nvt_ber": (C.c_int, [VP, c_dp, c_ip, c_i64, VP, VP, C.c_double, c_i64, C.c_double, C.c_double,
                                    C.c_double, VP, VP, VP, VP, VP, c_i64, c_dp]),
    "nepmi_nhc_init": (C.c_int, [VP, c_i64, C.c_double, C.c_double, C.c_double, VP]),
    "nepmi_nhc_half_step": (C.c_int, [VP, c_i64, C.c_double, C.c_double, VP, VP, VP]),
    "nepmi_run_nvt_nhc": (C.c_int, [VP, c_dp, c_ip, c_i64, VP, VP, C.c_double, c_i64, C.c_double, C.c_double,
                                    C.c_double, VP, VP, VP, VP, VP, c_i64, c_dp]),
    "nepmi_engine_reset_thermostat": (C.c_int, [VP]),
    "nepmi_bdp_seed": (C.c_int, [VP, C.c_uint64]),
    "nepmi_bdp_scale": (C.c_int, [VP, c_i64, C.c_double, C.c_double, VP, VP]),
    "nepmi_run_nvt_bdp": (C.c_int, [VP, c_dp, c_ip, c_i64, VP, VP, C.c_double, c_i64, C.c_double, C.c_double,
                                    C.c_double, VP, VP, VP, VP, VP, c_i64, c_dp]),
    "nepmi_lan_seed": (C.c_int, [VP, C.c_int]),
    "nepmi_lan_half_step": (C.c_int, [VP, c_i64, C.c_double, C.c_double, VP, VP]),
    "nepmi_run_nvt_lan": (C.c_int, [VP, c_dp, c_ip, c_i64, VP, VP, C.c_double, c_i64, C.c_double, C.c_double,
                                    C.c_double, VP, VP, VP, VP, VP, c_i64, c_dp]),
    "nepmi_run_nvt_bao": (C.c_int, [VP, c_dp, c_ip, c_i64, VP, VP, C.c_double, c_i64, C.c_double, C.c_double,
                                    C.c_double, VP, VP, VP, VP, VP, c_i64, c_dp]),
    "nepmi_neighbors_export": (C.c_int, [VP, C.c_int, VP, VP, c_i64]),
    "nepmi_descriptors_export": (C.c_int, [VP, VP, VP]),
    "nepmi_engine_stats": (C.c_int, [VP, C.c_int, C.POINTER(NepmiStats)]),
    "nepmi_engine_set_timing": (C.c_int, [VP, C.c_int]),
    "nepmi_engine_set_force_form": (C.c_int, [VP, C.c_int]),
    "nepmi_engine_set_option": (C.c_int, [VP, C.c_char_p, C.c_double]),
    "nepmi_engine_set_virial_mode": (C.c_int, [VP, C.c_int]),
    "nepmi_engine_describe": (C.c_int, [VP, C.c_char_p, C.c_int]),
    "nepmi_engine_set_temperature": (C.c_int, [VP, C.c_double]),
    "nepmi_transport_rccl_id": (C.c_int, [C.c_char_p]),
    "nepmi_transport_rccl": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(NepmiTransport)]),
    "nepmi_transport_tcp": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(NepmiTransport)]),
    "nepmi_transport_rccl_stats": (C.c_int, [C.POINTER(NepmiTransport), C.c_int, C.c_int, C.POINTER(NepmiRcclStats)]),
    "nepmi_transport_destroy": (None, [C.POINTER(NepmiTransport)]),
    "nepmi_dist_create": (VP, [VP, C.POINTER(NepmiTransport), c_dp, c_ip, c_ip, VP]),
    "nepmi_dist_destroy": (None, [VP]),
    "nepmi_dist_setup": (C.c_int, [VP, c_i64, VP, VP, VP, VP, VP]),
    "nepmi_dist_compute": (C.c_int, [VP]),
    "nepmi_dist_run": (C.c_int, [VP, C.c_int, C.c_double, c_i64, C.c_double, C.c_double, C.c_double, c_i64, c_dp]),
    "nepmi_dist_thermo": (C.c_int, [VP, c_dp]),
    "nepmi_dist_bdp_seed": (C.c_int, [VP, C.c_uint64]),
    "nepmi_dist_lan_seed": (C.c_int, [VP, C.c_int]),
    "nepmi_dist_set_overlap": (C.c_int, [VP, C.c_int]),
    "nepmi_dist_set_ghost_mode": (C.c_int, [VP, C.c_int]),
    "nepmi_dist_get_info": (C.c_int, [VP, C.POINTER(NepmiDistInfo)]),
    "nepmi_dist_info_bytes": (C.c_int, []),
    "nepmi_dist_num_overlapped_reverse": (C.c_int64, [VP]),
    "nepmi_dist_gather_owned": (C.c_int, [VP, VP, VP, VP, VP, VP, VP]),
    "nepmi_dist_gather_global": (C.c_int, [VP, C.c_int, VP, VP, VP, VP, VP]),
    "nepmi_dist_reset_thermostat": (C.c_int, [VP]),
    "nepmi_dist_engine": (VP, [VP]),
}


def bind(lib):
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


class NepmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("nepmi error %d: %s" % (code, msg))
        self.code = code


def check(lib, status):
    if status < 0:
        raise NepmiError(status, lib.nepmi_last_error().decode())
    return status
