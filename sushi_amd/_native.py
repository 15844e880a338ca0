"""ctypes binding of libsushi_hip.so (C ABI: include/sushi_hip.h).

There is no CPU fallback: if the library is missing or a call fails, the product path raises.
"""
import ctypes
import os
import re

import numpy as np

from .common import SushiError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SUSHI_HIP_LIB") or os.path.join(_HERE, "lib", "libsushi_hip.so")   # env: dev A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "sushi_hip.h")

# dtype codes / paths (include/sushi_hip.h)
U8, F32 = 0, 1
PATH_FFT, PATH_DIRECT = 0, 1
METHOD_SQDIFF_NORMED, METHOD_CCOEFF_NORMED = 0, 1       # cv2.TM_SQDIFF_NORMED + argmin (wav.py:185-186) | cv2.TM_CCOEFF_NORMED + argmax
METHODS = {"sqdiff_normed": METHOD_SQDIFF_NORMED, "ccoeff_normed": METHOD_CCOEFF_NORMED}
VIEW_XC, VIEW_S1, VIEW_S2, VIEW_UREL, VIEW_BASE, VIEW_SPECTRA, VIEW_USREL, VIEW_BASE1, VIEW_COARSE, VIEW_SPECTRA_LOW, \
    VIEW_ZNORM_REST = range(11)

ABI_VERSION = 13
NSTAGES = 6
STAGE_NAMES = ("tspec", "mac", "ifft", "refine", "finish", "bound")
STAGE_KERNELS = {"tspec": "tspec_kernel", "mac": "mac_kernel", "ifft": "mac_list_kernel+mac_rows_kernel+ifft_kernel", "refine": "refine_kernel",
                 "finish": "collect_kernel+exact_tiles_kernel+unpack_keys_kernel", "bound": "bound_low_kernel|bound_kernel"}
EXCLUSION = {"auto": 0, "always": 1, "never": 2, "band": 3, "whole": 4}        # SUSHI_HIP_EXCLUDE_*
WS_TSPEC, WS_Y, WS_TSPEC_LOW, WS_Y_LOW = range(4)                               # SUSHI_HIP_WS_*
BOUND_MODEL = {"worst_case": 0, "statistical": 1}                              # SUSHI_HIP_BOUND_*
# every kernel a stage's HIP-event span covers (profiles/pmc_traffic.json is keyed by kernel)
STAGE_KERNEL_SETS = {"tspec": ("tspec_kernel",), "mac": ("mac_kernel", "mac_long_kernel"),
                     "ifft": ("ifft_kernel", "ifft_list_kernel", "pilot_kernel", "survivor_kernel", "mac_list_kernel", "mac_rows_kernel",
                              "bound_low_exact_kernel", "slb_list_kernel", "survivor2_kernel"),
                     "refine": ("refine_kernel",),
                     "finish": ("collect_kernel", "exact_tiles_kernel"), "bound": ("bound_kernel", "bound_low_kernel", "slb_kernel")}

# struct SushiHipRequest, 24 bytes
REQUEST_DTYPE = np.dtype([("tmpl_off", "<i8"), ("win_start", "<i8"), ("tmpl_len", "<i4"), ("n_pos", "<i4")], align=True)
assert REQUEST_DTYPE.itemsize == 24


class BatchInfo(ctypes.Structure):
    _fields_ = [("n_search", ctypes.c_int32), ("path", ctypes.c_int32), ("variant", ctypes.c_int32),
                ("sub_batches", ctypes.c_int32), ("direct_tiles", ctypes.c_int64), ("fft_pairs", ctypes.c_int64),
                ("fft_segments", ctypes.c_int64), ("workspace_bytes", ctypes.c_uint64), ("mem_bytes", ctypes.c_uint64),
                ("flops", ctypes.c_double), ("algorithmic_bytes", ctypes.c_double), ("lanes", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class BatchDiag(ctypes.Structure):
    _fields_ = [("flagged", ctypes.c_int32), ("all_positions", ctypes.c_int32), ("tiles_dense", ctypes.c_int64),
                ("tiles_sparse", ctypes.c_int64), ("candidates", ctypes.c_int64), ("max_bound_ratio", ctypes.c_float),
                ("max_bound_ratio_noncandidate", ctypes.c_float), ("audited", ctypes.c_int64),
                ("pairs_transformed", ctypes.c_int64), ("excluded_audited", ctypes.c_int64),
                ("max_slb_ratio_excluded", ctypes.c_float), ("slb_violations", ctypes.c_int32), ("band", ctypes.c_int32),
                ("suspended", ctypes.c_int32), ("band_votes", ctypes.c_int32 * 2), ("second_look_audited", ctypes.c_int64)]


_lib = None


class NativeError(SushiError):
    pass


def declared_symbols():
    """Every function include/sushi_hip.h declares (used by the CPU-side export test)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    return sorted(set(re.findall(r"SUSHI_HIP_API\s+[\w\s\*]+?\b(sushi_hip_\w+)\s*\(", text)))


def lib():
    """Load (once) and type the library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError("libsushi_hip.so is not built (%s); run `python -m sushi_amd.build` -- "
                          "there is no CPU fallback for the matching path" % LIB_PATH)
    # PyTorch-ROCm ships its own HIP runtime (same SONAME as /opt/rocm's).  Whichever is mapped first serves the
    # whole process, and torch cannot see the GPU through the other one: torch first, then our library.
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    vp, i64, ci, dbl, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
    i32, u32, cf = ctypes.c_int32, ctypes.c_uint32, ctypes.c_float
    pvp = ctypes.POINTER(vp)
    L.sushi_hip_abi_version.restype = ci
    L.sushi_hip_strerror.restype = ctypes.c_char_p
    L.sushi_hip_strerror.argtypes = [ci]
    L.sushi_hip_device_ok.restype = ci
    L.sushi_hip_device_prepare.restype = ci
    L.sushi_hip_fft_size.restype = ci
    L.sushi_hip_fft_block.restype = ci
    L.sushi_hip_fft_slot_of_bin.restype = ci
    L.sushi_hip_fft_slot_of_bin.argtypes = [ci]
    L.sushi_hip_fft_low_slot_of_bin.restype = ci
    L.sushi_hip_fft_low_slot_of_bin.argtypes = [ci]
    L.sushi_hip_centre.restype = dbl
    L.sushi_hip_centre.argtypes = [ci]
    L.sushi_hip_stream_bytes.restype = sz
    L.sushi_hip_stream_bytes.argtypes = [i64, ci, ci]
    L.sushi_hip_stream_spectra_bytes.restype = sz
    L.sushi_hip_stream_spectra_bytes.argtypes = [i64]
    L.sushi_hip_stream_create.restype = ci
    L.sushi_hip_stream_create.argtypes = [vp, ci, i64, ci, vp, sz, vp, pvp]
    L.sushi_hip_stream_add_spectra.restype = ci
    L.sushi_hip_stream_add_spectra.argtypes = [vp, vp, sz, vp]
    L.sushi_hip_stream_view.restype = ci
    L.sushi_hip_stream_view.argtypes = [vp, ci, pvp, ctypes.POINTER(sz)]
    L.sushi_hip_stream_destroy.restype = None
    L.sushi_hip_stream_destroy.argtypes = [vp]
    L.sushi_hip_batch_bytes.restype = sz
    L.sushi_hip_batch_bytes.argtypes = [vp, ci, ci, ci, sz]
    L.sushi_hip_batch_create.restype = ci
    L.sushi_hip_batch_create.argtypes = [vp, vp, vp, ci, ci, ci, sz, vp, sz, vp, pvp]
    L.sushi_hip_batch_reset.restype = ci
    L.sushi_hip_batch_reset.argtypes = [vp, vp, ci, vp]
    L.sushi_hip_batch_info.restype = ci
    L.sushi_hip_batch_info.argtypes = [vp, ctypes.POINTER(BatchInfo)]
    L.sushi_hip_batch_set_method.restype = ci
    L.sushi_hip_batch_set_method.argtypes = [vp, ci]
    L.sushi_hip_batch_run.restype = ci
    L.sushi_hip_batch_run.argtypes = [vp, dbl, vp, vp, vp]
    L.sushi_hip_batch_diagnostics.restype = ci
    L.sushi_hip_batch_diagnostics.argtypes = [vp, ctypes.POINTER(BatchDiag), vp, vp]
    L.sushi_hip_batch_set_packed_output.restype = ci
    L.sushi_hip_batch_set_packed_output.argtypes = [vp, vp]
    L.sushi_hip_batch_set_early_output.restype = ci
    L.sushi_hip_batch_set_early_output.argtypes = [vp, vp]
    L.sushi_hip_batch_set_exclusion.restype = ci
    L.sushi_hip_batch_set_exclusion.argtypes = [vp, ci]
    L.sushi_hip_batch_set_bound_model.restype = ci
    L.sushi_hip_batch_set_bound_model.argtypes = [vp, ci]
    L.sushi_hip_batch_pair_bounds.restype = ci
    L.sushi_hip_batch_pair_bounds.argtypes = [vp, vp, vp, ctypes.POINTER(i64)]
    L.sushi_hip_batch_workspace_view.restype = ci
    L.sushi_hip_batch_workspace_view.argtypes = [vp, ci, pvp, ctypes.POINTER(sz)]
    L.sushi_hip_batch_destroy.restype = None
    L.sushi_hip_batch_destroy.argtypes = [vp]
    L.sushi_hip_fft_layout.restype = ci
    L.sushi_hip_fft_layout.argtypes = [i64, i32, i32, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.sushi_hip_load_decode.restype = ci
    L.sushi_hip_load_decode.argtypes = [vp, i64, i32, i32, vp, vp]
    L.sushi_hip_load_resample.restype = ci
    L.sushi_hip_load_resample.argtypes = [vp, i64, i32, i32, dbl, i64, i32, i32, dbl, i64, i64, vp, vp]
    L.sushi_hip_load_histogram.restype = ci
    L.sushi_hip_load_histogram.argtypes = [vp, i64, ci, u32, u32, ci, vp, vp]
    L.sushi_hip_load_normalise.restype = ci
    L.sushi_hip_load_normalise.argtypes = [vp, i64, cf, cf, cf, vp, vp]
    L.sushi_hip_profile_begin.restype = ci
    L.sushi_hip_profile_end.restype = ci
    L.sushi_hip_profile_end.argtypes = [vp, ci, ctypes.POINTER(ci)]
    if L.sushi_hip_abi_version() != ABI_VERSION:
        raise NativeError("libsushi_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = lib().sushi_hip_strerror(rc).decode()
        raise NativeError("%s failed: %s (%d)" % (what, msg, rc))


def fft_layout(win_start, n_pos, tmpl_len):
    """(block pairs, pattern segments) of one request on the FFT path (sushi_hip_fft_layout)."""
    a, b = ctypes.c_int32(0), ctypes.c_int32(0)
    check(lib().sushi_hip_fft_layout(int(win_start), int(n_pos), int(tmpl_len), ctypes.byref(a), ctypes.byref(b)),
          "sushi_hip_fft_layout")
    return a.value, b.value


def profile_begin():
    check(lib().sushi_hip_profile_begin(), "sushi_hip_profile_begin")


def profile_end(max_calls):
    """-> float32[n_calls, NSTAGES] milliseconds per stage of every FFT-path call since profile_begin."""
    buf = np.zeros((max(1, max_calls), NSTAGES), np.float32)
    n = ctypes.c_int(0)
    check(lib().sushi_hip_profile_end(buf.ctypes.data, int(max_calls), ctypes.byref(n)), "sushi_hip_profile_end")
    return buf[:n.value]
