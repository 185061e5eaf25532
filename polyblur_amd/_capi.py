"""ctypes binding of include/polyblur_hip.h (libpolyblur_hip.so).

This is the whole Python<->native boundary: plain pointers and sizes, no torch types.
The library is loaded lazily; a missing library or a missing GPU raises -- there is no
CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

PB_KSIZE = 25
PB_KSIZE_MAX = 49          # ker_size above PB_KSIZE: the large-kernel pass (the records stay 25 x 25)
PB_MAX_ANGLES = 13
PB_MAX_INTERP = 64
PB_MAX_PHASES = 175

PB_F32, PB_F16, PB_U8 = 0, 1, 2
PB_WRAP, PB_ZERO = 0, 1
PB_PREFILTER_NONE, PB_PREFILTER_BILATERAL, PB_PREFILTER_DOMAIN_TRANSFORM, PB_PREFILTER_NORMALIZED_CONVOLUTION = 0, 1, 2, 3
PB_SUPPORT_FULL, PB_SUPPORT_ADAPTIVE = 0, 1
PB_DENSE_STENCIL, PB_DENSE_AUTO = 0, 1
PB_DENSE_MIN_PHASES = 16
PB_COMM_CHUNK_AUTO = -1      # pb_comm_set_chunk: pb_comm_default_chunk's rule (0 = image by image, the default)

STATUS = {0: "PB_OK", -1: "PB_ERR_BADARG", -2: "PB_ERR_UNSUPPORTED", -3: "PB_ERR_HIP", -4: "PB_ERR_NOMEM"}

# every symbol include/polyblur_hip.h declares (tests check the library exports them all)
SYMBOLS = [
    "pb_version", "pb_create", "pb_destroy", "pb_set_stream", "pb_synchronize", "pb_last_error_string",
    "pb_default_options", "pb_workspace_bytes", "pb_malloc", "pb_free", "pb_memcpy_h2d", "pb_memcpy_d2h",
    "pb_polyblur_batch", "pb_estimate_blur", "pb_make_kernels", "pb_set_kernels", "pb_fourier_gradients",
    "pb_inverse_filter", "pb_convolve2d", "pb_edgetaper", "pb_halo_mask", "pb_dt_recursive_filter",
    "pb_bilateral5", "pb_time_inner_loop", "pb_profile_begin", "pb_profile_end", "pb_extract_patches",
    "pb_overlap_add", "pb_u8_deinterleave", "pb_u8_interleave", "pb_dt_normalized_convolution",
    "pb_fft_length_supported", "pb_make_separable_kernels", "pb_set_dense_eval",
    "pb_body_selection",
    "pb_comm_shard", "pb_comm_unique_id", "pb_comm_init", "pb_comm_destroy", "pb_comm_scatter", "pb_comm_gather",
    "pb_comm_deblur_from_root", "pb_comm_plan_steps", "pb_comm_plan", "pb_comm_set_chunk", "pb_comm_default_chunk",
    "pb_comm_plan_steps_chunked", "pb_comm_plan_chunked",
]
PROF_TAGS = ["conv", "gray", "grad_rows", "grad_cols", "params", "halo", "prefilter", "other", "conv_fused", "conv_fft"]


class pb_options(C.Structure):
    _fields_ = [
        ("n_iter", C.c_int32), ("c", C.c_float), ("b", C.c_float), ("alpha", C.c_float), ("beta", C.c_float),
        ("sigma_s", C.c_float), ("sigma_r", C.c_float), ("q", C.c_float), ("n_angles", C.c_int32),
        ("n_interpolated_angles", C.c_int32), ("remove_halo", C.c_int32), ("edgetaping", C.c_int32),
        ("prefilter", C.c_int32), ("discard_saturation", C.c_int32), ("boundary", C.c_int32),
        ("support", C.c_int32), ("force_theta_deg", C.c_float), ("separable_approx", C.c_int32), ("half_temporaries", C.c_int32), ("ker_size", C.c_int32),
    ]


class pb_blur_info(C.Structure):
    _fields_ = [
        ("gray_min", C.c_float), ("gray_max", C.c_float), ("mags", C.c_float * PB_MAX_ANGLES),
        ("interp", C.c_float * PB_MAX_INTERP), ("i_min", C.c_int32), ("theta", C.c_float),
        ("sigma", C.c_float), ("rho", C.c_float), ("separable", C.c_int32), ("radius", C.c_int32),
        ("kernel", C.c_float * (PB_KSIZE * PB_KSIZE)), ("kx", C.c_float * PB_KSIZE), ("ky", C.c_float * PB_KSIZE),
        ("acorr_y", C.c_float * PB_KSIZE), ("acorr_x", C.c_float * PB_KSIZE),
        ("gtaps", C.c_float * ((PB_KSIZE + 1) * 32)), ("gtaps_odd", C.c_float * ((PB_KSIZE + 1) * 32)),
        ("xt_first", C.c_int32), ("xt_exact_rank1", C.c_int32), ("xt_g1", C.c_float * PB_KSIZE), ("xt_m", C.c_int32 * PB_KSIZE),
        ("xt_wa", C.c_float * PB_KSIZE), ("xt_wb", C.c_float * PB_KSIZE),
        ("nphase", C.c_int32 * 3), ("phase", C.c_int32 * (PB_MAX_PHASES + 9)),
    ]


INFO_DTYPE = np.dtype([
    ("gray_min", "<f4"), ("gray_max", "<f4"), ("mags", "<f4", (PB_MAX_ANGLES,)), ("interp", "<f4", (PB_MAX_INTERP,)),
    ("i_min", "<i4"), ("theta", "<f4"), ("sigma", "<f4"), ("rho", "<f4"), ("separable", "<i4"), ("radius", "<i4"),
    ("kernel", "<f4", (PB_KSIZE, PB_KSIZE)), ("kx", "<f4", (PB_KSIZE,)), ("ky", "<f4", (PB_KSIZE,)),
    ("acorr_y", "<f4", (PB_KSIZE,)), ("acorr_x", "<f4", (PB_KSIZE,)), ("gtaps", "<f4", (PB_KSIZE + 1, 32)),
    ("gtaps_odd", "<f4", (PB_KSIZE + 1, 32)), ("xt_first", "<i4"), ("xt_exact_rank1", "<i4"), ("xt_g1", "<f4", (PB_KSIZE,)),
    ("xt_m", "<i4", (PB_KSIZE,)), ("xt_wa", "<f4", (PB_KSIZE,)), ("xt_wb", "<f4", (PB_KSIZE,)), ("nphase", "<i4", (3,)), ("phase", "<i4", (PB_MAX_PHASES + 9,)),
])
assert INFO_DTYPE.itemsize == C.sizeof(pb_blur_info)


class PolyblurHipError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()


def library_path() -> str:
    env = os.environ.get("POLYBLUR_HIP_LIB")
    if env:
        return env
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libpolyblur_hip.so")


def load_library():
    """dlopen libpolyblur_hip.so and declare the prototypes.  Raises if it is missing."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.exists(path):
            raise PolyblurHipError(
                "libpolyblur_hip.so not found at %s -- build it with `python -m polyblur_amd.build` "
                "(there is no CPU fallback)" % path)
        lib = C.CDLL(path)
        vp, ci, cf, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
        fp = C.POINTER(C.c_float)
        proto = {
            "pb_version": (ci, []),
            "pb_create": (ci, [C.POINTER(vp), ci, vp]),
            "pb_destroy": (ci, [vp]),
            "pb_set_stream": (ci, [vp, vp]),
            "pb_synchronize": (ci, [vp]),
            "pb_last_error_string": (C.c_char_p, [vp]),
            "pb_default_options": (None, [C.POINTER(pb_options)]),
            "pb_workspace_bytes": (sz, [vp]),
            "pb_set_dense_eval": (ci, [vp, ci, ci]),
            "pb_malloc": (ci, [vp, C.POINTER(vp), sz]),
            "pb_free": (ci, [vp, vp]),
            "pb_memcpy_h2d": (ci, [vp, vp, vp, sz]),
            "pb_memcpy_d2h": (ci, [vp, vp, vp, sz]),
            "pb_polyblur_batch": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, C.POINTER(pb_options), vp]),
            "pb_u8_deinterleave": (ci, [vp, vp, vp, ci, ci, ci, ci]),
            "pb_u8_interleave": (ci, [vp, vp, vp, ci, ci, ci, ci]),
            "pb_estimate_blur": (ci, [vp, vp, ci, ci, ci, ci, ci, C.POINTER(pb_options), vp]),
            "pb_make_kernels": (ci, [vp, ci, fp, fp, fp, ci, vp]),
            "pb_set_kernels": (ci, [vp, ci, fp, ci, vp]),
            "pb_fourier_gradients": (ci, [vp, vp, ci, ci, ci, vp, vp]),
            "pb_inverse_filter": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, vp, cf, cf, ci, ci, ci, vp, vp]),
            "pb_convolve2d": (ci, [vp, vp, vp, ci, ci, ci, ci, vp, ci]),
            "pb_edgetaper": (ci, [vp, vp, vp, ci, ci, ci, ci, vp, ci]),
            "pb_halo_mask": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci]),
            "pb_dt_recursive_filter": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, cf, cf, ci]),
            "pb_dt_normalized_convolution": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, cf, cf, ci]),
            "pb_bilateral5": (ci, [vp, vp, vp, ci, ci, ci, ci, ci]),
            "pb_time_inner_loop": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, vp, cf, cf, ci, ci, fp]),
            "pb_extract_patches": (ci, [vp, vp, vp] + [ci] * 15),
            "pb_overlap_add": (ci, [vp, vp, vp] + [ci] * 13 + [vp, vp]),
            "pb_fft_length_supported": (ci, [ci]),
            "pb_make_separable_kernels": (ci, [vp, ci, vp, vp, ci, ci]),
            "pb_body_selection": (ci, [vp, ci, C.POINTER(ci), ci]),
            "pb_profile_begin": (ci, [vp]),
            "pb_profile_end": (ci, [vp, fp, C.POINTER(ci)]),
            "pb_comm_shard": (ci, [ci, ci, ci, C.POINTER(ci), C.POINTER(ci)]),
            "pb_comm_unique_id": (ci, [C.c_char_p]),
            "pb_comm_init": (ci, [C.POINTER(vp), vp, ci, ci, C.c_char_p]),
            "pb_comm_destroy": (ci, [vp]),
            "pb_comm_scatter": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci]),
            "pb_comm_gather": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci]),
            "pb_comm_deblur_from_root": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, C.POINTER(pb_options), ci]),
            "pb_comm_plan_steps": (ci, [ci, ci, ci]),
            "pb_comm_plan": (ci, [ci, ci, ci, ci, ci, C.POINTER(ci), C.POINTER(ci)]),
            "pb_comm_set_chunk": (ci, [vp, ci]),
            "pb_comm_default_chunk": (ci, [ci, ci, ci]),
            "pb_comm_plan_steps_chunked": (ci, [ci, ci, ci, ci]),
            "pb_comm_plan_chunked": (ci, [ci, ci, ci, ci, ci, ci, C.POINTER(ci), C.POINTER(ci)]),
        }
        for name in SYMBOLS:
            fn = getattr(lib, name)           # AttributeError if the library does not export it
            fn.restype, fn.argtypes = proto[name]
        _lib = lib
        return lib
