"""ctypes binding of librxmd_hip.so (the C ABI of include/rxmd_hip.h).

There is no CPU implementation behind this module: if the shared library (hand-written HIP
kernels compiled for gfx950) is missing, importing fails loudly with the build command.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("RXMD_HIP_LIB") or os.path.join(HERE, "librxmd_hip.so")     # RXMD_HIP_LIB: A/B builds of the same library


class RxmdConfig(C.Structure):
    _fields_ = [("ffield_path", C.c_char_p), ("lattice", C.c_double * 6), ("vprocs", C.c_int * 3), ("myid", C.c_int),
                ("isQEq", C.c_int), ("NMAXQEq", C.c_int), ("QEq_tol", C.c_double), ("qstep", C.c_int), ("dt_fs", C.c_double),
                ("Lex_fqs", C.c_double), ("Lex_k", C.c_double), ("nbuffer", C.c_int), ("maxneighbs", C.c_int),
                ("maxneighbs10", C.c_int), ("device", C.c_int), ("qeq_mode", C.c_int), ("lg", C.c_int),
                ("pqeq_path", C.c_char_p), ("efield_dir", C.c_int), ("reserved1", C.c_int), ("efield_strength", C.c_double)]


class RxmdStats(C.Structure):
    _fields_ = [("natoms", C.c_int), ("nghost_qeq", C.c_int), ("nghost_force", C.c_int), ("nnz10", C.c_longlong), ("nbonds", C.c_longlong),
                ("max_n10", C.c_int), ("max_nb", C.c_int), ("qeq_iters_last", C.c_int), ("qeq_iters_total", C.c_longlong), ("qeq_calls", C.c_longlong),
                ("ms_qeq", C.c_double), ("ms_qeq_list", C.c_double), ("ms_qeq_spmv", C.c_double), ("ms_force", C.c_double), ("ms_lists", C.c_double),
                ("ms_bo", C.c_double), ("ms_nonbond", C.c_double), ("ms_bonded", C.c_double), ("ms_step_total", C.c_double),
                ("spmv_launches", C.c_longlong), ("n10_stride", C.c_int), ("nbuffer", C.c_int), ("cells10", C.c_int * 3), ("cells3", C.c_int * 3),
                ("n_boundary_rows", C.c_int), ("spmv_noop_launches", C.c_int), ("reserved", C.c_int * 6),
                ("ms_ghost_build", C.c_double), ("ms_migrate", C.c_double), ("ms_halo", C.c_double), ("ms_halo_exposed", C.c_double),
                ("ms_allreduce", C.c_double), ("ms_fold", C.c_double), ("halo_calls", C.c_longlong), ("allreduce_calls", C.c_longlong),
                ("ms_k_list10", C.c_double), ("ms_k_nonbond", C.c_double), ("ms_k_e3b", C.c_double), ("ms_k_e4b", C.c_double), ("ms_k_ehb", C.c_double),
                ("ms_k_bondorder", C.c_double), ("ms_k_assemble", C.c_double), ("ms_k_winbuild", C.c_double),
                ("win_groups", C.c_int), ("win_max_units", C.c_int), ("win_in_use", C.c_int), ("reserved2", C.c_int),
                ("place_ms_first", C.c_double), ("place_ms_kept", C.c_double),
                ("place_total_ms", C.c_double), ("place_bytes_held", C.c_double), ("place_draws", C.c_int),
                ("spmv_nstep", C.c_int), ("spmv_var", C.c_int), ("reserved3", C.c_int), ("ms_k_blist", C.c_double),
                ("ms_bond_exposed", C.c_double), ("bond_overlap", C.c_int), ("spmv_launches_timed", C.c_longlong), ("timer_pairs_dropped", C.c_int)]

    def asdict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        d.pop("reserved"); d.pop("reserved2"); d.pop("reserved3")
        return d


EXCHANGE_FN = C.CFUNCTYPE(C.c_longlong, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_longlong)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)


class RxmdCommOps(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("exchange", EXCHANGE_FN), ("allreduce_sum", ALLREDUCE_FN), ("exchange_known", EXCHANGE_FN)]


# every symbol include/rxmd_hip.h declares: (name, restype, argtypes)
H = C.c_void_p
PD = C.c_void_p
SYMBOLS = [
    ("rxmd_hip_default_config", None, [C.POINTER(RxmdConfig)]),
    ("rxmd_hip_create", C.c_int, [C.POINTER(RxmdConfig), C.POINTER(H)]),
    ("rxmd_hip_destroy", C.c_int, [H]),
    ("rxmd_hip_last_error", C.c_char_p, [H]),
    ("rxmd_hip_set_atoms_rxff", C.c_int, [H, C.c_int, PD]),
    ("rxmd_hip_get_atoms_rxff", C.c_int, [H, PD, C.c_int]),
    ("rxmd_hip_get_atoms", C.c_int, [H, C.c_int, PD, PD, PD, PD, PD, PD]),
    ("rxmd_hip_set_charges", C.c_int, [H, C.c_int, PD]),
    ("rxmd_hip_set_velocities", C.c_int, [H, C.c_int, PD]),
    ("rxmd_hip_get_shells", C.c_int, [H, PD, C.c_int]),
    ("rxmd_hip_get_bonds", C.c_int, [H, C.c_int, C.c_int, C.c_void_p, C.c_void_p, PD]),
    ("rxmd_hip_set_shells", C.c_int, [H, C.c_int, PD]),
    ("rxmd_hip_qeq", C.c_int, [H, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    ("rxmd_hip_force", C.c_int, [H, PD]),
    ("rxmd_hip_step", C.c_int, [H, C.c_int]),
    ("rxmd_hip_get_energy", C.c_int, [H, C.POINTER(C.c_double), C.POINTER(C.c_double), PD, PD]),
    ("rxmd_hip_QEq", C.c_int, [H, C.c_int, C.c_int, PD, PD, PD]),
    ("rxmd_hip_FORCE", C.c_int, [H, C.c_int, C.c_int, PD, PD, PD, PD, PD]),
    ("rxmd_hip_get_stats", C.c_int, [H, C.POINTER(RxmdStats)]),
    ("rxmd_hip_reset_timers", C.c_int, [H]),
    ("rxmd_hip_PQEq", C.c_int, [H, C.c_int, C.c_int, PD, PD, PD, PD]),
    ("rxmd_hip_FORCE_pqeq", C.c_int, [H, C.c_int, C.c_int, PD, PD, PD, PD, PD, PD]),
    ("rxmd_hip_put_lex", C.c_int, [H, C.c_int, PD, PD]),
    ("rxmd_hip_get_lex", C.c_int, [H, C.c_int, PD, PD]),
    ("rxmd_hip_minimise", C.c_int, [H, C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    ("rxmd_hip_last_qeq_iters", C.c_int, [H]),
    ("rxmd_hip_thermostat", C.c_int, [H, C.c_int, C.c_double, C.c_double, C.c_double]),
    ("rxmd_hip_rccl_unique_id", C.c_int, [C.c_char_p]),
    ("rxmd_hip_comm_init_rccl", C.c_int, [H, C.c_char_p, C.c_int, C.c_int]),
    ("rxmd_hip_set_qeq_mode", C.c_int, [H, C.c_int]),
    ("rxmd_hip_get_table", C.c_int, [H, C.c_int, PD, C.c_int]),
    ("rxmd_hip_get_cutoffs", C.c_int, [H, PD, C.c_int, C.POINTER(C.c_double)]),
    ("rxmd_hip_debug_get", C.c_int, [H, C.c_int, PD, C.c_int]),
    ("rxmd_hip_set_comm", C.c_int, [H, C.POINTER(RxmdCommOps)]),
    ("rxmd_hip_copy_to_host", C.c_int, [PD, PD, C.c_longlong]),
    ("rxmd_hip_copy_to_device", C.c_int, [PD, PD, C.c_longlong]),
    ("rxmd_hip_device_count", C.c_int, []),
    ("rxmd_hip_set_exchange_buffers", C.c_int, [H, PD, PD, C.c_longlong]),
    ("rxmd_host_comm_selftest", C.c_int, [C.POINTER(RxmdCommOps), C.c_int, C.c_int]),
    ("rxmd_host_geninit", C.c_longlong, [C.c_char_p, C.c_int, C.c_char_p, PD, PD, PD, PD, C.c_int, PD, C.c_longlong, PD]),
    ("rxmd_host_read_rxff", C.c_longlong, [C.c_char_p, C.c_int, PD, PD, PD, C.c_longlong]),
    ("rxmd_host_ffield_table", C.c_int, [C.c_char_p, PD, C.c_int, PD, C.c_longlong]),
    ("rxmd_host_ffield_lg", C.c_int, [C.c_int]),
    ("rxmd_host_describe_options", C.c_int, [C.c_char_p, C.c_int]),
    ("rxmd_hip_has_device_code", C.c_int, []),
]


def describe_options():
    """the environment switches of the library as the markdown rows of README.md (rxmd_amd/csrc/options.def); no GPU needed"""
    lib = load()
    n = lib.rxmd_host_describe_options(None, 0)
    buf = C.create_string_buffer(n + 1)
    lib.rxmd_host_describe_options(buf, n + 1)
    return buf.value.decode()

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            "rxmd_amd: %s is missing. This package has no CPU path: build the HIP kernels first with\n"
            "    python -c 'import __graft_entry__ as g; g.build()'   (or: make -C rxmd_amd/csrc)" % SO_PATH)
    lib = C.CDLL(SO_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError here = ABI drift between header and library
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
