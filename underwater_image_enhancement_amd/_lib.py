"""ctypes binding of libuwie.so (include/uwie.h).  No CPU fallback: a missing library is an error."""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libuwie.so")
CSRC = os.path.join(_HERE, "csrc")

SURFACE_SIX, SURFACE_DICT = 0, 1
INTER_F64, INTER_FX32, INTER_F32T = 0, 1, 2  # uwie_params.inter_dtype
DICT_STRATEGIES = {
    "strong_dehazing": 0,
    "medium_dehazing": 1,
    "light_enhancement": 2,
    "clahe_enhancement": 3,
    "histogram_equalization": 4,
}
CAST_KINDS = ("normal", "greenish", "bluish")


class UwieError(RuntimeError):
    pass


class UwieParams(ctypes.Structure):
    """Mirror of ``struct uwie_params`` (include/uwie.h)."""

    _fields_ = [
        ("surface", ctypes.c_int32),
        ("strategy", ctypes.c_int32),
        ("cast_correct", ctypes.c_int32),
        ("forced_cast", ctypes.c_int32),
        ("gray_shift", ctypes.c_int32),
        ("min_size", ctypes.c_int32),
        ("omega", ctypes.c_double),
        ("gf_ksize", ctypes.c_int32),
        ("gf_eps", ctypes.c_double),
        ("L_low", ctypes.c_double),
        ("L_high", ctypes.c_double),
        ("wb_percentile", ctypes.c_double),
        ("clip_limit", ctypes.c_double),
        ("tiles_x", ctypes.c_int32),
        ("tiles_y", ctypes.c_int32),
        ("gamma", ctypes.c_double),
        ("apply_gamma", ctypes.c_int32),
        ("gf_exact", ctypes.c_int32),
        ("inter_dtype", ctypes.c_int32),
    ]


# name -> argument types (return type is int unless listed in _RESTYPES)
_VP, _SZ, _I, _D = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_double
_PP = ctypes.POINTER(UwieParams)
SIGNATURES = {
    "uwie_last_error": [],
    "uwie_version": [],
    "uwie_create": [_I, ctypes.POINTER(_VP)],
    "uwie_destroy": [_VP],
    "uwie_device_status": [_VP, _VP, ctypes.POINTER(ctypes.c_uint32)],
    "uwie_profile_enable": [_VP, _I],
    "uwie_profile_filter": [_VP, ctypes.c_char_p],
    "uwie_profile_collect": [_VP],
    "uwie_profile_row": [_VP, _I, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(_D), ctypes.POINTER(_I)],
    "uwie_params_init": [_PP, _I, _I],
    "uwie_set_tuning": [_VP, ctypes.c_char_p, _I],
    "uwie_get_tuning": [_VP, ctypes.c_char_p, ctypes.POINTER(_I)],
    "uwie_workspace_bytes": [_I, _I, _I, _PP],
    "uwie_workspace_bytes_ctx": [_VP, _I, _I, _I, _PP],
    "uwie_workspace_bytes_all": [_I, _I, _I, _VP],
    "uwie_workspace_bytes_float": [_I, _I, _I, _PP, _I],
    "uwie_enhance_f32": [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _PP, _VP, _SZ, _VP],
    "uwie_enhance_f64": [_VP, _VP, _VP, _VP, _I, _I, _I, _PP, _VP, _SZ, _VP],
    "uwie_cast_classify_f32": [_VP, _VP, _I, _I, _I, _VP, _VP, _VP],
    "uwie_color_correct_f32": [_VP, _VP, _VP, _VP, _I, _I, _I, _VP],
    "uwie_guided_plan": [_I, _I, _I, _I, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)],
    "uwie_enhance_u8": [_VP, _VP, _VP, _VP, _I, _I, _I, _PP, _VP, _SZ, _VP],
    "uwie_enhance_u8_f64": [_VP, _VP, _VP, _VP, _I, _I, _I, _PP, _VP, _SZ, _VP],
    "uwie_enhance_all_u8": [_VP, _VP, _VP, _VP, _I, _I, _I, _VP, _VP, _SZ, _VP],
    "uwie_workspace_bytes_select": [_I, _I, _I, _VP, _I, _I],
    "uwie_select_best_u8": [_VP, _VP, _I, _I, _I, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _SZ, _VP],
    "uwie_diff_enhance_f32": [_VP, _VP, _VP, _I, _I, _I, _I, _VP, _I, _VP, _SZ, _VP],
    "uwie_extract_features_u8": [_VP, _VP, _VP, _I, _I, _I, _VP, _SZ, _VP],
    "uwie_quality_scores": [_VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _SZ, _VP],
    "uwie_cast_classify": [_VP, _VP, _I, _I, _I, _VP, _VP, _VP, _SZ, _VP],
    "uwie_normalise_correct": [_VP, _VP, _VP, _VP, _I, _I, _I, _VP],
    "uwie_atmospheric_light": [_VP, _VP, _VP, _I, _I, _I, _PP, _VP, _VP, _VP, _SZ, _VP],
    "uwie_transmission_init": [_VP, _VP, _VP, _VP, _I, _I, _I, _PP, _VP, _VP, _VP],
    "uwie_box_filter_f64": [_VP, _VP, _VP, _I, _I, _I, _I, _VP, _SZ, _VP],
    "uwie_guided_filter": [_VP, _VP, _VP, _I, _I, _I, _I, _D, _I, _VP, _VP, _SZ, _VP],
    "uwie_restore": [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP, _VP],
    "uwie_percentiles_f32": [_VP, _VP, _I, _I, _I, ctypes.POINTER(_D), _I, _VP, _VP, _SZ, _VP],
    "uwie_stretch_f32": [_VP, _VP, _VP, _I, _I, _I, _D, _D, _VP, _SZ, _VP],
    "uwie_gamma_f32": [_VP, _VP, _VP, _SZ, _D, _I, _VP],
    "uwie_clahe_f32": [_VP, _VP, _VP, _I, _I, _I, _D, _I, _I, _VP, _SZ, _VP],
    "uwie_rgb2gray_u8": [_VP, _VP, _VP, _SZ, _I, _VP],
    "uwie_rgb2lab_u8": [_VP, _VP, _VP, _SZ, _VP],
    "uwie_lab2rgb_u8": [_VP, _VP, _VP, _SZ, _VP],
    "uwie_clahe_u8": [_VP, _VP, _VP, _I, _I, _I, _D, _I, _I, _VP, _SZ, _VP],
    "uwie_canny_u8": [_VP, _VP, _VP, _I, _I, _I, _I, _I, _VP, _SZ, _VP],
    "uwie_equalize_hist_u8": [_VP, _VP, _VP, _I, _I, _I, _VP, _SZ, _VP],
}
_RESTYPES = {
    "uwie_last_error": ctypes.c_char_p,
    "uwie_version": ctypes.c_char_p,
    "uwie_destroy": None,
    "uwie_workspace_bytes": ctypes.c_size_t,
    "uwie_workspace_bytes_ctx": ctypes.c_size_t,
    "uwie_workspace_bytes_all": ctypes.c_size_t,
    "uwie_workspace_bytes_float": ctypes.c_size_t,
    "uwie_workspace_bytes_select": ctypes.c_size_t,
}

_lib = None


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-j8"])
    return LIB_PATH


def load() -> ctypes.CDLL:
    """Load libuwie.so; raises UwieError when it has not been built (there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UwieError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(the enhancement path has no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the .so does not match include/uwie.h
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, ctypes.c_int)
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().uwie_last_error()
        raise UwieError(f"libuwie error {rc}: {msg.decode() if msg else '?'}")
