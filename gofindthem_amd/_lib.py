"""ctypes binding of libgft.so (include/gft.h).  Fails loudly when the library is missing: there is no
Python/CPU implementation of the hot path in this package."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GFT_LIBRARY") or os.path.join(HERE, "libgft.so")   # GFT_LIBRARY: another build (tools/asan_host.sh)

GFT_OK, GFT_E_INVALID, GFT_E_NOT_BUILT, GFT_E_HIP, GFT_E_UNSUPPORTED, GFT_E_PARSE, GFT_E_ENGINE = 0, -1, -2, -3, -4, -5, -6
GFT_E_NOMEM, GFT_E_INTERNAL, GFT_W_NO_RCCL = -7, -8, 1
GFT_POS_START, GFT_POS_END = 0, 1
GFT_FOLD_ASCII = 1
GFT_SCAN_UNIQUE = 2
GFT_POS_RUNES = 4
OP_UNIT, OP_AND, OP_OR, OP_NOT, OP_INORD = 1, 2, 3, 4, 5
INORD_FLAG = 1 << 27


class GftMatches(C.Structure):
    _fields_ = [("n_docs", C.c_uint64), ("n_matches", C.c_uint64), ("match_off", C.c_void_p),
                ("term_id", C.c_void_p), ("pos", C.c_void_p)]


class GftExtra(C.Structure):
    _fields_ = [("off", C.c_void_p), ("slot", C.c_void_p), ("pos", C.c_void_p)]


# every symbol include/gft.h declares: (restype, argtypes)
_vp, _u32, _u64, _i = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
SYMBOLS = {
    "gft_engine_create": (_i, [C.POINTER(_vp), _i]),
    "gft_engine_destroy": (None, [_vp]),
    "gft_engine_create_multi": (_i, [C.POINTER(_vp), _vp, _i]),
    "gft_n_devices": (_i, [_vp]),
    "gft_gather_mode": (C.c_char_p, [_vp]),
    "gft_build_info": (C.c_char_p, []),
    "gft_device_engine": (_vp, [_vp, _i]),
    "gft_split_docs": (_i, [_vp, _vp, _u64, _vp]),
    "gft_process_device_multi": (_i, [_vp, _vp, _vp, _vp, _u32, _vp]),
    "gft_last_error": (C.c_char_p, [_vp]),
    "gft_set_stream": (_i, [_vp, _vp]),
    "gft_set_cu_margin": (_i, [_vp, C.c_uint32]),
    "gft_build": (_i, [_vp, _vp, _vp, _u32, _u32]),
    "gft_export_tables": (_i, [_vp, _vp, _u64, C.POINTER(_u64)]),
    "gft_import_tables": (_i, [_vp, C.c_char_p, _u64]),
    "gft_n_terms": (_u32, [_vp]),
    "gft_n_states": (_u32, [_vp]),
    "gft_last_nonascii": (_i, [_vp]),
    "gft_scan_kernel": (C.c_char_p, [_vp]),
    "gft_debug_scan5_filter": (_i, [_vp, _vp, _u32, _vp, _u32, _u32, _u32, _u32, _vp, _vp, C.POINTER(_u32)]),
    "gft_term": (_i, [_vp, _u32, C.POINTER(_vp), C.POINTER(_u32)]),
    "gft_term_id": (C.c_int64, [_vp, _vp, _u32]),
    "gft_scan": (_i, [_vp, _vp, _vp, _u64, _u32, C.POINTER(GftMatches)]),
    "gft_scan_device": (_i, [_vp, _vp, _vp, _u64, _u32, C.POINTER(GftMatches)]),
    "gft_set_programs": (_i, [_vp, _vp, _vp, _u32, _u32]),
    "gft_n_exprs": (_u32, [_vp]),
    "gft_n_host_exprs": (_u32, [_vp]),
    "gft_process": (_i, [_vp, _vp, _vp, _u64, _u32, C.POINTER(GftExtra), _vp]),
    "gft_process_again": (_i, [_vp, _u64, C.POINTER(GftExtra), _vp]),
    "gft_process_device": (_i, [_vp, _vp, _vp, _u64, _u32, C.POINTER(GftExtra), _vp]),
    "gft_process_device_begin": (_i, [_vp, _vp, _vp, _u64, _u32, C.POINTER(GftExtra), _vp]),
    "gft_process_device_end": (_i, [_vp]),
    "gft_finder_create": (_i, [C.POINTER(_vp), _i, _i]),
    "gft_finder_create_multi": (_i, [C.POINTER(_vp), _i, _vp, _i]),
    "gft_finder_destroy": (None, [_vp]),
    "gft_finder_last_error": (C.c_char_p, [_vp]),
    "gft_finder_engine": (_vp, [_vp]),
    "gft_finder_set_substring_engine": (_i, [_vp, _vp, _vp, _vp]),
    "gft_finder_set_regex_engine": (_i, [_vp, _vp, _vp, _vp]),
    "gft_finder_add_expression": (_i, [_vp, C.c_char_p, _u64, C.c_char_p, _u64]),
    "gft_finder_n_expressions": (_u32, [_vp]),
    "gft_finder_n_literals": (_u32, [_vp, _i]),
    "gft_finder_literal": (_i, [_vp, _i, _u32, C.POINTER(_vp), C.POINTER(_u32)]),
    "gft_finder_expression": (_i, [_vp, _u32, C.POINTER(_vp), C.POINTER(_u32), C.POINTER(_vp), C.POINTER(_u32),
                                   C.POINTER(_vp), C.POINTER(_u32)]),
    "gft_finder_force_build": (_i, [_vp]),
    "gft_finder_process_text": (_i, [_vp, C.c_char_p, _u64, _vp, _u32, C.POINTER(_u32)]),
    "gft_finder_process_texts": (_i, [_vp, _vp, _vp, _u64, _vp]),
    "gft_finder_last_regex_docs": (_u64, [_vp]),
    "gft_finder_process_device": (_i, [_vp, _vp, _vp, _u64, _vp]),
    "gft_finder_process_device_begin": (_i, [_vp, _vp, _vp, _u64, _vp]),
    "gft_finder_process_device_end": (_i, [_vp]),
    "gft_finder_debug_add_literal": (_i, [_vp, _i, C.c_char_p, _u32]),
    "gft_finder_debug_set_updated": (_i, [_vp, _i, _i]),
    "gft_finder_debug_get_updated": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "gft_group_create": (_i, [C.POINTER(_vp), _vp]),
    "gft_group_destroy": (None, [_vp]),
    "gft_group_last_error": (C.c_char_p, [_vp]),
    "gft_group_add_rule": (_i, [_vp, C.c_char_p, _u64, C.c_char_p, _u64]),
    "gft_group_state": (_i, [_vp, _vp, _u64, C.POINTER(_u64)]),
    "gft_group_process_jsons": (_i, [_vp, _vp, _vp, _u64, C.c_char_p, _u64, C.c_char_p, _u64, _i, _vp, _u64, C.POINTER(_u64)]),
    "gft_group_last_result": (_i, [_vp, _vp, _u64, C.POINTER(_u64)]),
    "gft_group_evaluate": (_i, [_vp, C.c_char_p, _u64, _vp, _u64, C.POINTER(_u64)]),
    "gft_group_last_batch": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "gft_group_dsl_parse": (_i, [C.c_char_p, _u64, _vp, _u64, C.POINTER(_u64)]),
    "gft_group_dsl_tokens": (_i, [C.c_char_p, _u64, _vp, _u64, C.POINTER(_u64)]),
    "gft_dsl_parse": (_i, [C.c_char_p, _u64, _i, _vp, _u64, C.POINTER(_u64)]),
    "gft_regex_required_literals": (_i, [C.c_char_p, _u64, _vp, _u64, C.POINTER(_u64)]),
    "gft_dsl_tokens": (_i, [C.c_char_p, _u64, _vp, _u64, C.POINTER(_u64)]),
    "gft_to_lower": (_i, [C.c_char_p, _u64, _vp, _u64, C.POINTER(_u64)]),
    "gft_debug_emulate_scan": (_i, [_vp, _vp, _u32, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _u64, C.POINTER(_u64)]),
    "gft_debug_eval_programs": (_i, [_vp, _vp, _u32, _u32, _vp, _vp, _vp]),
    "gft_debug_host_solve": (_i, [_vp, _u64, _vp, _vp, _vp, _u32, C.POINTER(_i)]),
    "gft_profile_enable": (_i, [_vp, _i]),
    "gft_profile_read": (_i, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_u64)]),
    "gft_profile_reset": (_i, [_vp]),
}

EMIT_FN = C.CFUNCTYPE(None, _vp, _vp, _u32, C.c_int64)
BUILD_FN = C.CFUNCTYPE(_i, _vp, _vp, _vp, _u32, _i, _vp, _u32)
FIND_FN = C.CFUNCTYPE(_i, _vp, _vp, _u64, EMIT_FN, _vp, _vp, _u32)

_LIB = None


def _mapped_hip_runtimes():
    """paths of the libamdhip64 copies mapped into this process"""
    try:
        with open("/proc/self/maps") as f:
            return sorted({ln.split()[-1] for ln in f if "libamdhip64.so" in ln})
    except OSError:
        return []


def _hip_runtime_first():
    """libgft.so links /opt/rocm's libamdhip64, PyTorch ships its own copy of the same soname.  Whichever is mapped first
    serves both -- as long as torch comes FIRST: with libgft.so loaded before torch the process has carried two HIP runtimes
    (the one abort in the records).  So a process that can import torch imports it before the library
    is mapped; one without torch has a single runtime anyway.  GFT_NO_TORCH_PRELOAD=1 skips this (a caller that never
    imports torch and does not want its start-up time)."""
    import sys
    if "torch" in sys.modules or os.environ.get("GFT_NO_TORCH_PRELOAD"):
        return
    import importlib.util
    if importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401


def _check_one_hip_runtime():
    libs = _mapped_hip_runtimes()
    if len(libs) > 1:
        import warnings
        warnings.warn("two HIP runtimes are mapped into this process (%s): device pointers of one are not valid in the "
                      "other -- import torch before gofindthem_amd loads libgft.so" % ", ".join(libs), RuntimeWarning)


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libgft.so is not built (run `python -m gofindthem_amd.build`); "
                               "gofindthem_amd has no CPU fallback for the hot path")
        _hip_runtime_first()
        L = C.CDLL(LIB_PATH)
        _check_one_hip_runtime()
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)            # AttributeError if a declared symbol is not exported
            f.restype, f.argtypes = res, args
        _LIB = L
    return _LIB
