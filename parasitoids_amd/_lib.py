"""ctypes binding of libparasitoid_hip.so (C ABI: include/parasitoid_hip.h).

Importing this module does not touch the GPU (fork-safe, like the reference's
lazy `import cuda_lib`, CalcSol.py:162).  `load()` raises ImportError when the
shared library is missing; there is no CPU fallback in this package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libparasitoid_hip.so')

PS_OK = 0
PS_ERR_NO_DEVICE = -1
PS_ERR_OOM = -2
PS_ERR_BAD_SHAPE = -3
PS_ERR_BAD_ARG = -4
PS_ERR_UNSUPPORTED = -5
PS_ERR_HIP = -6
PS_ERR_HPROB_BOUNDS = -7
PS_ERR_PMF_NEGATIVE = -8
PS_ERR_FLIGHT_PROB = -9
PS_ERR_STATE = -10
PS_ERR_EMPTY = -11

MODE_EXACT = 0
MODE_FAST = 1
MODE_FOLD = 2
MODE_AUTO = 3

REC_CHAIN, REC_BACK, REC_STATE, REC_WSUM = 0, 1, 2, 3


class DayStats(C.Structure):
    _fields_ = [('nnz', C.c_int64), ('sum', C.c_double), ('delta', C.c_double),
                ('padmax', C.c_double), ('flag', C.c_int32), ('pad_', C.c_int32)]


class HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('libparasitoid_hip error %d: %s' % (code, msg))
        self.code = code


_I32P = C.POINTER(C.c_int32)
_I64P = C.POINTER(C.c_int64)
_F64P = C.POINTER(C.c_double)
_VP = C.c_void_p

# name -> (restype, argtypes); every symbol include/parasitoid_hip.h declares
SIGNATURES = {
    'ps_version': (C.c_int, []),
    'ps_device_count': (C.c_int, []),
    'ps_last_error': (C.c_char_p, []),
    'ps_device_info': (C.c_int, [C.c_int, C.c_char_p, C.c_int, _I32P, _I64P]),
    'ps_solver_create': (C.c_int, [C.POINTER(_VP), C.c_int, C.c_int, C.c_int, C.c_int]),
    'ps_solver_destroy': (C.c_int, [_VP]),
    'ps_solver_info': (C.c_int, [_VP, _I32P, _I32P, _I32P, _I32P]),
    'ps_solver_sync': (C.c_int, [_VP]),
    'ps_solver_set_option': (C.c_int, [_VP, C.c_char_p, C.c_double]),
    'ps_solver_get_option': (C.c_int, [_VP, C.c_char_p, _F64P]),
    'ps_solver_set_state_coo': (C.c_int, [_VP, _I32P, _I32P, _F64P, C.c_int64]),
    'ps_solver_fftconv2_coo': (C.c_int, [_VP, _I32P, _I32P, _F64P, C.c_int64, C.c_int]),
    'ps_solver_get_cursol': (C.c_int, [_VP, C.c_double, C.c_double, C.c_int, C.POINTER(DayStats)]),
    'ps_solver_back_solve': (C.c_int, [_VP, C.c_int, _I64P, _I32P, _I32P, _F64P, C.c_double,
                                       C.c_double, C.POINTER(DayStats)]),
    'ps_chain_set_kernels': (C.c_int, [_VP, C.c_int, _I64P, _I32P, _I32P, _I32P, _F64P]),
    'ps_chain_run': (C.c_int, [_VP, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]),
    'ps_chain_stats': (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(DayStats)]),
    'ps_chain_run_release': (C.c_int, [_VP, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _F64P,
                                       C.POINTER(C.c_int)]),
    'ps_record_stats': (C.c_int, [_VP, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                  C.POINTER(DayStats)]),
    'ps_solver_retarget': (C.c_int, [_VP, C.c_int]),
    'ps_fast_size': (C.c_int, [C.c_int, C.c_int]),
    'ps_solver_kernels_direct': (C.c_int, [_VP]),
    'ps_solver_pipeline': (C.c_int, [_VP]),
    'ps_solver_auto_info': (C.c_int, [_VP, _I32P, _I32P]),
    'ps_solver_auto_route': (C.c_int, [_VP, C.c_int, C.c_int, _I32P]),
    'ps_record_fetch_coo': (C.c_int, [_VP, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                      C.c_double, _I32P, _I32P, _F64P, C.c_int64, _I64P]),
    'ps_record_fetch_csr': (C.c_int, [_VP, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                      C.c_double, _I32P, _I32P, _F64P, C.c_int64, _I64P]),
    'ps_record_fetch_dense': (C.c_int, [_VP, C.c_int, C.c_int, _F64P]),
    'ps_record_gather': (C.c_int, [_VP, C.c_int, C.c_int, C.c_int64, _I32P, _I32P, C.c_double,
                                   C.c_double, _F64P]),
    'ps_record_gather_multi': (C.c_int, [_VP, C.c_int, _I32P, _I32P, C.c_int64, _I32P, _I32P, C.c_double,
                                          C.c_double, _F64P]),
    'ps_weighted_sum': (C.c_int, [_VP, C.c_int, _I32P, _I32P, _F64P]),
    'ps_prof_enable': (C.c_int, [_VP, C.c_int]),
    'ps_prof_read': (C.c_int, [_VP, C.c_int, _F64P, _I64P]),
    'ps_prof_read_days': (C.c_int, [_VP, C.c_int, _I64P]),
    'ps_prof_read_launches': (C.c_int, [_VP, C.c_int, _I64P]),
    'ps_chain_block_prefix': (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    'ps_chain_block_finish': (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_double, C.c_double,
                                        C.c_int, C.POINTER(C.c_int)]),
    'ps_device_copy': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    'ps_prof_read_owner': (C.c_int, [_VP, C.c_int, C.c_int, _F64P, _I64P, _I64P, _I64P]),
    'ps_solver_owner_fft': (C.c_int, [_VP, C.c_int]),
    'ps_solver_get_spectrum': (C.c_int, [_VP, _F64P]),
    'ps_solver_set_spectrum': (C.c_int, [_VP, _F64P]),
    'ps_model_create': (C.c_int, [C.POINTER(_VP), C.c_int]),
    'ps_model_destroy': (C.c_int, [_VP]),
    'ps_model_set_option': (C.c_int, [_VP, C.c_char_p, C.c_double]),
    'ps_model_set_wind': (C.c_int, [_VP, _F64P, _I32P, C.c_int, C.c_int, C.c_int]),
    'ps_model_prob_mass': (C.c_int, [_VP, C.c_int, _I32P, _F64P, _F64P, _F64P, _F64P, C.c_double,
                                     C.c_int, C.c_double, C.c_int, _I32P, _I64P, _I32P, _I32P]),
    'ps_model_fetch_coo': (C.c_int, [_VP, C.c_int, _I32P, _I32P, _F64P, C.c_int64]),
    'ps_model_fetch_debug': (C.c_int, [_VP, C.c_int, _F64P, _I32P, _F64P, _F64P]),
    'ps_model_hflight': (C.c_int, [_VP, C.c_int, _F64P, _F64P]),
    'ps_model_mvn_cdf_values': (C.c_int, [_VP, C.c_double, C.c_double, C.c_double, C.c_double,
                                          C.c_double, C.c_double, _I32P, _F64P, C.c_int64]),
    'ps_chain_set_kernels_from_model': (C.c_int, [_VP, _VP, C.c_int, C.c_int]),
    'ps_solver_set_state_from_model': (C.c_int, [_VP, _VP, C.c_int]),
    'ps_model_export_device': (C.c_int, [_VP, C.c_int, C.c_int, _VP, _VP, _VP, C.c_int64]),
    'ps_chain_set_kernels_device': (C.c_int, [_VP, C.c_int, _I64P, _I32P, _VP, _VP, _VP]),
    'ps_solver_set_state_device': (C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, C.c_int]),
}

_lib = None


def load():
    """Load the shared library (no HIP call is made).  ImportError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libparasitoid_hip.so not found at %s; build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` or '
            '`make -C parasitoids_amd/csrc`' % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise ImportError('cannot load %s: %s' % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != PS_OK:
        msg = load().ps_last_error()
        raise HipError(rc, msg.decode() if msg else '')


def require_device():
    """ImportError if no usable GPU (mirrors the reference: `import cuda_lib`
    failing makes CalcSol fall back, CalcSol.py:161-172)."""
    lib = load()
    n = lib.ps_device_count()
    if n <= 0:
        msg = lib.ps_last_error()
        raise ImportError('no MI355X/HIP device available: %s' % (msg.decode() if msg else ''))
    return n


def default_device():
    """LOCAL_RANK-aware default device (one process per GPU)."""
    return int(os.environ.get('PARASITOID_DEVICE', os.environ.get('LOCAL_RANK', '0')))


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def p_i32(a):
    return a.ctypes.data_as(_I32P)


def p_i64(a):
    return a.ctypes.data_as(_I64P)


def p_f64(a):
    return a.ctypes.data_as(_F64P)
