"""ctypes binding of libpips_hip.so (include/pips_hip.h).

The library is the product: there is no eager/PyTorch fallback.  ``load()`` raises if
the shared object is missing or does not export every declared symbol.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpips_hip.so")

c_void_p, c_int, c_size_t, c_float = C.c_void_p, C.c_int, C.c_size_t, C.c_float
fp = c_void_p  # device float pointer

# name -> (restype, argtypes); mirrors include/pips_hip.h declaration by declaration
SIGNATURES = {
    "pips_last_error": (C.c_char_p, []),
    "pips_abi_version": (c_int, []),
    "pips_weight_arena_bytes": (c_size_t, []),
    "pips_repack_weights": (c_int, [C.POINTER(c_void_p), c_int, c_void_p, c_void_p]),
    "pips_repack_weights_ex": (c_int, [C.POINTER(c_void_p), c_int, c_void_p, c_int, c_void_p]),
    "pips_workspace_bytes": (c_size_t, [c_int] * 6),
    "pips_forward": (c_int, [c_void_p, fp, fp, fp, fp, fp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                             c_void_p, c_size_t, fp, fp, fp, c_void_p]),
    "pips_forward_ce": (c_int, [c_void_p, fp, fp, fp, fp, fp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_void_p, c_size_t, fp, fp, fp, fp, fp, c_void_p, c_size_t, c_void_p]),
    "pips_track_ce": (c_int, [c_void_p, fp, c_int, c_int, c_int, c_int, fp, fp, fp, c_void_p, fp, c_int, c_int, c_int,
                              c_int, c_void_p, c_size_t, fp, fp, fp, fp, fp, c_void_p, c_size_t, c_void_p]),
    "pips_score_map_workspace_bytes": (c_size_t, [c_int] * 4),
    "pips_score_map_prepare": (c_int, [fp, c_int, c_int, c_int, c_int, fp, c_void_p]),
    "pips_score_map_terms": (c_int, [fp, c_int, c_int, c_int, c_int, fp, c_int, fp, fp, c_void_p]),
    "pips_track_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pips_weight_arena_bytes_s": (c_size_t, [c_int]),
    "pips_repack_weights_s": (c_int, [C.POINTER(c_void_p), c_int, c_void_p, c_int, c_int, c_void_p]),
    "pips_delta_stride": (c_int, [c_int]),
    "pips_mixer_workspace_bytes_s": (c_size_t, [c_int, c_int]),
    "pips_mixer_fwd_s": (c_int, [c_void_p, fp, c_int, c_int, c_int, fp, c_void_p, c_size_t, c_void_p]),
    "pips_track_workspace_bytes_s": (c_size_t, [c_int, c_int, c_int]),
    "pips_track_s": (c_int, [c_void_p, fp, c_int, c_int, c_int, c_int, fp, fp, fp, c_void_p, fp, c_int, c_int, c_int,
                             c_int, c_int, c_void_p, c_size_t, fp, fp, fp, fp, fp, c_void_p, c_size_t, c_void_p]),
    "pips_track": (c_int, [c_void_p, fp, c_int, c_int, c_int, c_int, fp, fp, fp, c_void_p, fp, c_int, c_int, c_int,
                           c_int, c_void_p, c_size_t, fp, fp, fp, c_void_p]),
    "pips_encoder_workspace_bytes": (c_size_t, [c_int] * 4),
    "pips_pyramid_floats": (c_size_t, [c_int] * 4),
    "pips_pyramid_offset": (c_size_t, [c_int] * 5),
    "pips_pyramid_mirror_offset": (c_size_t, [c_int] * 4),
    "pips_pyramid_mirror": (c_int, [fp, c_int, c_int, c_int, c_int, c_void_p]),
    "pips_mixer_input_build_ex": (c_int, [fp, c_int, c_int, c_int, c_int, fp, fp, fp, c_int, c_void_p, c_int, fp, c_void_p]),
    "pips_encoder_fwd": (c_int, [c_void_p, fp, c_int, c_int, c_int, c_int, fp, c_void_p, c_size_t, c_void_p]),
    "pips_encoder_fwd_bf16": (c_int, [c_void_p, fp, c_int, c_int, c_int, c_int, fp, c_void_p, c_size_t, c_void_p]),
    "pips_encoder_fwd_ex": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, fp, c_void_p, c_size_t,
                                    c_void_p]),
    "pips_resize_frames": (c_int, [c_void_p, c_int, c_int, c_int, c_int, fp, c_int, c_int, c_void_p]),
    "pips_point_sample": (c_int, [fp, c_int, c_int, c_int, c_int, fp, c_int, fp, c_void_p]),
    "pips_mixer_input_build": (c_int, [fp, c_int, c_int, c_int, c_int, fp, fp, fp, c_int, fp, c_void_p]),
    "pips_gather_scratch_bytes": (c_size_t, [c_int] * 4),
    "pips_mixer_input_build_tiled": (c_int, [fp, c_int, c_int, c_int, c_int, fp, fp, fp, c_int, fp, c_void_p, c_size_t,
                                             c_void_p]),
    "pips_mixer_input_build_tiled_timed": (c_int, [fp, c_int, c_int, c_int, c_int, fp, fp, fp, c_int, fp, c_void_p,
                                                   c_size_t, c_void_p, C.POINTER(c_float)]),
    "pips_gather_route": (c_int, [c_int] * 5),
    "pips_mixer_input_build_tiled_ex": (c_int, [fp, c_int, c_int, c_int, c_int, fp, fp, fp, c_int, c_int, fp, c_void_p, c_size_t,
                                                c_void_p, C.POINTER(c_float)]),
    "pips_mixer_workspace_bytes": (c_size_t, [c_int]),
    "pips_mixer_fwd": (c_int, [c_void_p, fp, c_int, fp, c_void_p, c_size_t, c_void_p]),
    "pips_mixer_fwd_bf16": (c_int, [c_void_p, fp, c_int, fp, c_void_p, c_size_t, c_void_p]),
    "pips_mixer_fwd_x3": (c_int, [c_void_p, fp, c_int, fp, c_void_p, c_size_t, c_void_p]),
    "pips_mixer_fwd_timed": (c_int, [c_void_p, fp, c_int, fp, c_void_p, c_size_t, c_void_p, C.POINTER(c_float)]),
    "pips_mixer_fwd_timed_ex": (c_int, [c_void_p, fp, c_int, c_int, fp, c_void_p, c_size_t, c_void_p,
                                        C.POINTER(c_float)]),
    "pips_mixer_gemm_train": (c_int, [c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_int, C.POINTER(c_float)]),
    "pips_state_update": (c_int, [c_void_p, fp, fp, fp, fp, c_int, c_int, c_float, fp, fp, c_void_p]),
    "pips_gemm_f32": (c_int, [fp, c_int, fp, fp, fp, c_int, c_int, c_int, c_int, c_int, fp, c_int, c_void_p]),
    "pips_conv_nhwc_f32": (c_int, [fp, c_int, c_int, c_int, c_int, fp, fp, c_int, c_int, c_int, c_int, fp, fp,
                                   C.POINTER(c_int), c_void_p]),
    "pips_gemm_bf16": (c_int, [c_void_p, c_int, c_int, c_void_p, fp, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, fp, c_int,
                       c_void_p]),
    "pips_gemm_bf16_route": (c_int, [c_int] * 6),
    "pips_gemm_f32_route": (c_int, [c_int] * 4),
    "pips_device_cus": (c_int, []),
    "pips_conv_nhwc_bf16": (c_int, [fp, c_int, c_int, c_int, c_int, c_void_p, fp, c_int, c_int, c_int, c_int, fp, fp,
                                    C.POINTER(c_int), c_void_p]),
    "pips_conv_nhwc_bf16_maps": (c_int, [c_void_p, fp, c_int, c_int, c_int, c_int, c_void_p, fp, c_int, c_int, c_int, c_int,
                                         c_void_p, c_int, fp, c_int, C.POINTER(c_int), c_void_p]),
    "pips_split_bf16x3": (c_int, [fp, c_size_t, c_void_p, c_void_p]),
    "pips_gemm_f32x3": (c_int, [fp, c_int, c_void_p, fp, fp, c_int, c_int, c_int, c_int, c_int, fp, c_int, c_void_p]),
    "pips_conv_nhwc_f32x3": (c_int, [fp, c_int, c_int, c_int, c_int, c_void_p, fp, c_int, c_int, c_int, c_int, fp, fp,
                                     C.POINTER(c_int), c_void_p]),
}

_lock = threading.Lock()
_lib = None


class PipsHipError(RuntimeError):
    pass


def use_library(path: str):
    """Bind to another build of the library (tuning / trace builds of tools/).  Must be called before the first
    ``load()``; the product never reads an environment variable for this -- tools/_tunelib.py does, for the tools."""
    global LIB_PATH
    if _lib is not None:
        raise PipsHipError("pips_amd._lib.use_library: the library is already loaded")
    LIB_PATH = path


def load():
    """Load (once) and return the ctypes handle.  ``import torch`` must have happened
    before so that the HIP runtime the library binds to (SONAME libamdhip64.so.7) is the
    copy torch already loaded -- one runtime per process, so torch streams are valid here."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise PipsHipError(
                f"{LIB_PATH} is missing: build it with `python -m pips_amd._build` "
                "(or __graft_entry__.build()).  pips_amd has no CPU/PyTorch fallback.")
        import torch  # noqa: F401  (loads torch's libamdhip64 first)

        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise PipsHipError(f"libpips_hip.so does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().pips_last_error()
        raise PipsHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else c_void_p(t.data_ptr())
