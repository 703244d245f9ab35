"""ctypes binding of libmlbp.so (include/mlbp.h).  The ONLY door from Python into the engine.

There is no CPU fallback: if the shared library is missing the import fails loudly, and device
entry points called on a machine without an MI355X return MLBP_ENODEVICE, raised as MlbpError.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, 'libmlbp.so')

MLBP_OK, MLBP_EINVAL, MLBP_EHIP, MLBP_ENODEVICE, MLBP_ENOMEM, MLBP_EUNSUPPORTED = 0, -1, -2, -3, -4, -5
OP_UNARY, OP_PAIR_TM, OP_PAIR_MT, OP_VAR = 0, 1, 2, 3
NORM_ZERO, NORM_UNIFORM = 0, 1


class MlbpError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, 'libmlbp error %d: %s' % (code, msg))
        self.code = code


class Topology(C.Structure):
    _fields_ = [('n_vars', C.c_int32), ('n_factors', C.c_int32),
                ('fac_nvars', C.POINTER(C.c_int32)), ('fac_var', C.POINTER(C.c_int32)),
                ('fac_dim', C.POINTER(C.c_int32)), ('var_fac_off', C.POINTER(C.c_int32)),
                ('var_fac', C.POINTER(C.c_int32))]


class GradientArgs(C.Structure):
    _fields_ = [('B', C.c_int32), ('X', C.c_int32), ('n_msgs', C.c_int32), ('P', C.c_int32), ('U', C.c_int32),
                ('F_ee', C.c_int32), ('F_ed', C.c_int32), ('Vde', C.c_int32), ('n_pair_tables', C.c_int32),
                ('n_unary_tables', C.c_int32), ('msgs', C.c_void_p), ('pair_tables', C.c_void_p),
                ('pair_tab', C.c_void_p), ('pair_c_slot', C.c_void_p), ('pair_r_slot', C.c_void_p),
                ('pair_phi', C.c_void_p), ('pair_label', C.c_void_p), ('unary_tables', C.c_void_p),
                ('unary_tab', C.c_void_p), ('unary_kind', C.c_void_p), ('unary_obs', C.c_void_p),
                ('unary_label', C.c_void_p), ('phi_en_en', C.c_void_p), ('phi_en_en_w1', C.c_void_p),
                ('phi_en_de', C.c_void_p), ('phi_en_en_t', C.c_void_p), ('phi_en_en_w1_t', C.c_void_p),
                ('phi_en_de_t', C.c_void_p), ('phi_en_en_p', C.c_void_p), ('phi_en_en_w1_p', C.c_void_p),
                ('grad_en_en', C.c_void_p), ('grad_en_de', C.c_void_p), ('flags', C.c_int32), ('pair_tab_host', C.c_void_p), ('unary_expect', C.c_void_p),
                ('pair_slots_host', C.c_void_p), ('workspace', C.c_void_p), ('workspace_bytes', C.c_int64)]


class PotentialsJob(C.Structure):
    _fields_ = [('phi', C.c_void_p), ('theta', C.c_void_p), ('pot', C.c_void_p), ('pot_t', C.c_void_p),
                ('theta_stride', C.c_int64), ('pot_stride', C.c_int64), ('pot_t_stride', C.c_int64),
                ('rows', C.c_int32), ('cols', C.c_int32), ('F', C.c_int32), ('reserved', C.c_int32),
                ('expect', C.c_void_p), ('expect_stride', C.c_int64)]


class PosteriorArgs(C.Structure):
    _fields_ = [('labels', C.c_void_p), ('out', C.c_void_p), ('sum_out', C.c_void_p)]


class SweepArgs(C.Structure):
    _fields_ = [('B', C.c_int32), ('X', C.c_int32), ('n_pair_tables', C.c_int32),
                ('n_unary_tables', C.c_int32), ('pair_tables', C.c_void_p), ('pair_tab', C.c_void_p),
                ('unary_tables', C.c_void_p), ('unary_tab', C.c_void_p), ('msgs', C.c_void_p),
                ('normalize_messages', C.c_int32), ('init_messages', C.c_int32), ('marginals', C.c_void_p), ('gradient', C.c_void_p),
                ('flags', C.c_int32), ('pair_tab_host', C.c_void_p), ('pair_tables_f32', C.c_void_p), ('posterior', C.c_void_p)]


SWEEP_SHARED_PAIR_TABLES = 1      # include/mlbp.h MLBP_SWEEP_*
SWEEP_NO_MESSAGE_WRITEBACK = 2
SWEEP_PAIR_TABLES_F32 = 4
SWEEP_DENSE_TABLES = 8
SWEEP_APPROX_INFERENCE = 16
SWEEP_SKIP_UNCHANGED = 32
APPROX_K = 100
GRADIENT_SHARED_PAIR_TABLES = 1
GRADIENT_APPROX_BELIEFS = 2


_i32p = C.POINTER(C.c_int32)
_vp = C.c_void_p
_i32, _i64 = C.c_int32, C.c_int64

# name -> (restype, argtypes); mirrors include/mlbp.h one to one (tests/test_abi.py checks that).
SIGNATURES = {
    'mlbp_version': (C.c_int, []),
    'mlbp_arch': (C.c_char_p, []),
    'mlbp_last_error': (C.c_char_p, []),
    'mlbp_device_count': (C.c_int, []),
    'mlbp_last_sweep_kernel': (C.c_int, []),
    'mlbp_last_sweep_fused_gradient': (C.c_int, []),
    'mlbp_has_loops': (C.c_int, [C.POINTER(Topology), _i32]),
    'mlbp_message_schedule': (C.c_int, [C.POINTER(Topology), _i32, _i32p, _i32]),
    'mlbp_message_slots': (C.c_int, [C.POINTER(Topology), _i32p, _i32p, _i32p, _i32p]),
    'mlbp_compile_sweep': (C.c_int, [C.POINTER(Topology), _i32, _i32p, _i32, _i32p, _i32, _i32p]),
    'mlbp_program_create': (C.c_int, [_i32p, _i32, _i32p, _i32, _i32p, _i32, _i32, _i32, _i32,
                                      C.POINTER(_vp)]),
    'mlbp_program_destroy': (C.c_int, [_vp]),
    'mlbp_program_plan': (C.c_int, [_i32p, _i32, _i32p, _i32, _i32p, _i32, _i32, _i32, _i32, _i32p]),
    'mlbp_program_reserve': (C.c_int, [_vp, _i32]),
    'mlbp_program_set_readout': (C.c_int, [_vp, _i32, _i32p, _i32p]),
    'mlbp_program_exact_count': (C.c_int, [_vp, _i32]),
    'mlbp_program_skippable_updates': (C.c_int, [_vp]),
    'mlbp_program_status': (C.c_int, [_vp]),
    'mlbp_set_sweep_variant': (C.c_int, [_i32]),
    'mlbp_sweep_f64': (C.c_int, [_vp, C.POINTER(SweepArgs), _vp]),
    'mlbp_sweep_groups_f64': (C.c_int, [C.POINTER(_vp), C.POINTER(SweepArgs), _i32, _vp]),
    'mlbp_init_messages_f64': (C.c_int, [_vp, _i64, _i32, _vp]),
    'mlbp_marginals_f64': (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    'mlbp_log_posterior_f64': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    'mlbp_log_posterior_sum_f64': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    'mlbp_log_posterior_groups_f64': (C.c_int, [_vp, _i32, _i64, _i32, _vp, _vp]),
    'mlbp_pair_beliefs_f64': (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    'mlbp_gradient_f64': (C.c_int, [C.POINTER(GradientArgs), _vp]),
    'mlbp_gradient_status': (C.c_int, []),
    'mlbp_gradient_workspace_bytes': (C.c_int64, [C.POINTER(GradientArgs)]),
    'mlbp_unary_expectations_f64': (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    'mlbp_patch_unary_tables_f64': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    'mlbp_patch_gradient_f64': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    'mlbp_sum_rows_f64': (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    'mlbp_sum_rows_cat_f64': (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i64, _i32, _vp, _vp]),
    'mlbp_select_sum_rows_cat_f64': (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i64, _vp, _vp, _i32, _vp, _vp]),
    'mlbp_segment_sum_rows_f64': (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp, _vp]),
    'mlbp_step_statistics_f64': (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i64, _vp, _vp, _vp]),
    'mlbp_dense_dot_f64': (C.c_int, [_i32, _i32, _i32, _i32, _vp, _i64, _i64, _i64, _vp, _i64, _i64,
                                     _i64, _vp, _i64, _i64, _vp]),
    'mlbp_pointwise_multiply_f64': (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp]),
    'mlbp_normalize_f64': (C.c_int, [_vp, _vp, _i32, _i64, _i32, _vp, _vp]),
    'mlbp_topk_f64': (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp]),
    'mlbp_topk_rows_f64': (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp]),
    'mlbp_sparse_vec_mat_dot_f64': (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _i32, _i32, _vp, _vp]),
    'mlbp_sparse_dot_f64': (C.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    'mlbp_sparse_pointwise_multiply_f64': (C.c_int, [_vp, _vp, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _vp]),
    'mlbp_sparse_normalize_f64': (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp]),
    'mlbp_potentials_f64': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    'mlbp_potentials_multi_f64': (C.c_int, [_vp, _i32, _i32, _vp]),
    'mlbp_log_f64': (C.c_int, [_vp, _vp, _i64, _vp]),
    'mlbp_observed_minus_f64': (C.c_int, [_vp, _i64, _i64, _vp, _vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libmlbp.so not found at %s.  Build it with `python -m macaronicusermodeling_amd.build` '
            '(hipcc, gfx950).  There is no CPU fallback.' % LIB_PATH)
    # torch bundles its own libamdhip64.so.7; libmlbp.so must share THAT runtime instance (device
    # pointers and streams cross the boundary), so torch's copy has to be the one already mapped
    # when the loader resolves libmlbp's NEEDED entry.  Loading in the other order leaves two HIP
    # runtimes in the process and ours sees no device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def last_error():
    return lib.mlbp_last_error().decode('utf-8', 'replace')


def check(rc):
    """Raises MlbpError for negative return codes; returns rc otherwise."""
    if rc < 0:
        raise MlbpError(rc, last_error())
    return rc


def i32ptr(arr):
    """numpy int32 C-contiguous array -> POINTER(c_int32)."""
    return arr.ctypes.data_as(_i32p)
