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

# dtype codes (include/sushi_hip.h)
U8, F32 = 0, 1
SQDIFF_NORMED = 0

ABI_VERSION = 5
NSTAGES = 5
STAGE_NAMES = ("tspec", "mac", "ifft", "refine", "finish")
STAGE_KERNELS = {"tspec": "tspec_kernel", "mac": "mac_kernel", "ifft": "ifft_kernel", "refine": "refine_kernel",
                 "finish": "exact_flagged_kernel|match_flagged_kernel+unpack_keys_kernel"}

# struct SushiHipSearch, 40 bytes
SEARCH_DTYPE = np.dtype([("tmpl_off", "<i8"), ("win_start", "<i8"), ("tmpl_len", "<i4"),
                         ("n_pos", "<i4"), ("first_tile", "<i4"), ("first_pair", "<i4"),
                         ("first_seg", "<i4"), ("reserved", "<i4")], align=True)
assert SEARCH_DTYPE.itemsize == 40

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
    L.sushi_hip_abi_version.restype = ci
    L.sushi_hip_strerror.restype = ctypes.c_char_p
    L.sushi_hip_strerror.argtypes = [ci]
    L.sushi_hip_device_ok.restype = ci
    L.sushi_hip_variant_count.restype = ci
    L.sushi_hip_variant_tile_positions.restype = ci
    L.sushi_hip_variant_tile_positions.argtypes = [ci]
    L.sushi_hip_prepare_base_bytes.restype = sz
    L.sushi_hip_prepare_base_bytes.argtypes = [i64]
    L.sushi_hip_centre.restype = dbl
    L.sushi_hip_centre.argtypes = [ci]
    L.sushi_hip_prepare_stream.restype = ci
    L.sushi_hip_prepare_stream.argtypes = [vp, ci, i64, vp, vp, vp, vp, vp, sz, vp]
    L.sushi_hip_match_batch.restype = ci
    L.sushi_hip_match_batch.argtypes = [vp, vp, vp, i64, vp, vp, vp, i64, dbl, ci, vp, ci, ci, ci, vp, vp, vp, vp]
    i32 = ctypes.c_int32
    L.sushi_hip_fft_hop.restype = ci
    L.sushi_hip_spectra_blocks.restype = i64
    L.sushi_hip_spectra_blocks.argtypes = [i64]
    L.sushi_hip_spectra_bytes.restype = sz
    L.sushi_hip_spectra_bytes.argtypes = [i64]
    L.sushi_hip_fft_layout.restype = ci
    L.sushi_hip_fft_layout.argtypes = [i64, i32, i32, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.sushi_hip_fft_workspace_bytes.restype = sz
    L.sushi_hip_fft_workspace_bytes.argtypes = [i64, i64, i64]
    L.sushi_hip_prepare_spectra.restype = ci
    L.sushi_hip_prepare_spectra.argtypes = [vp, ci, i64, vp, sz, vp]
    L.sushi_hip_match_batch_fft.restype = ci
    L.sushi_hip_match_batch_fft.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, i64, vp, vp, ci, ci, vp, vp, ci, dbl,
                                            vp, sz, vp, vp, vp, vp, vp, vp]
    L.sushi_hip_fft_sub_batches.restype = ci
    L.sushi_hip_fft_sub_batches.argtypes = [vp, ci, sz]
    L.sushi_hip_fft_pair_order.restype = ci
    L.sushi_hip_fft_pair_order.argtypes = [vp, ci, sz, vp, i64]
    u32, cf = ctypes.c_uint32, ctypes.c_float
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


def variant_tiles():
    L = lib()
    return [L.sushi_hip_variant_tile_positions(v) for v in range(L.sushi_hip_variant_count())]


def fft_layout(win_start, n_pos, tmpl_len):
    """Vectorised twin of sushi_hip_fft_layout (csrc/sushi_common.hpp fft_layout): block pairs and
    template segments per search.  sushi_hip_match_batch_fft re-derives and checks the running sums."""
    hop = lib().sushi_hip_fft_hop()
    w = np.asarray(win_start, dtype=np.int64)
    p = np.asarray(n_pos, dtype=np.int64)
    m = np.asarray(tmpl_len, dtype=np.int64)
    k0 = w // hop
    kl = (w + p - 1) // hop
    return (kl - k0 + 2) // 2, (m + hop - 1) // hop


def profile_begin():
    check(lib().sushi_hip_profile_begin(), "sushi_hip_profile_begin")


def profile_end(max_calls):
    """-> float32[n_calls, NSTAGES] milliseconds per stage of every FFT-path call since profile_begin."""
    buf = np.zeros((max(1, max_calls), NSTAGES), np.float32)
    n = ctypes.c_int(0)
    check(lib().sushi_hip_profile_end(buf.ctypes.data, int(max_calls), ctypes.byref(n)), "sushi_hip_profile_end")
    return buf[:n.value]
