"""Build libsushi_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Two translation units, compiled separately (the MFMA kernel alone takes ~2 min) and linked:
  csrc/sushi_hip.hip  direct MFMA kernel, stream preparation, exact refinement   (-ffp-contract=off)
  csrc/sushi_fft.hip  overlap-save FFT path
  csrc/sushi_load.hip WavStream load pipeline (decimate / pad / median clip / scale / quantise)  (-ffp-contract=off)
"""
import math
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "sushi_hip.h")
LIB_DIR = os.path.join(_HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB = os.path.join(LIB_DIR, "libsushi_hip.so")
TWIDDLE_INC = os.path.join(CSRC, "_gen_twiddle16384.inc")

COMMON_DEPS = [HEADER, os.path.join(CSRC, "sushi_common.hpp"), os.path.join(CSRC, "sushi_internal.hpp")]
# -ffp-contract=off for sushi_hip.hip: its float64 epilogue restates cv2's operation order; a fused
# a*b-c*d would round differently from the reference (the hot loop is MFMA builtins, unaffected).
UNITS = [
    ("sushi_hip", ["-ffp-contract=off"], []),
    ("sushi_load", ["-ffp-contract=off"], []),      # NumPy's float32 operation order, no fused multiply-add
    # -fno-slp-vectorize: the SLP pass packs the complex MACs into v_pk_fma_f32 and pays for it in
    # register shuffles (v_mov / accvgpr traffic); plain v_fma_f32 already issues at the f32 peak rate.
    ("sushi_fft", ["-fno-slp-vectorize"],
     [os.path.join(CSRC, "fft_core.hpp"), os.path.join(CSRC, "mac_core.hpp"), TWIDDLE_INC, os.path.join(CSRC, "_gen_dft16_f16.inc"),
      # the translation unit's parts, by stage (included inside its anonymous namespace)
      os.path.join(CSRC, "sushi_fft_store.inc"), os.path.join(CSRC, "sushi_fft_spectra.inc"), os.path.join(CSRC, "sushi_fft_mac.inc"),
      os.path.join(CSRC, "sushi_fft_ifft.inc"), os.path.join(CSRC, "sushi_fft_bound.inc"), os.path.join(CSRC, "sushi_fft_collect.inc"),
      os.path.join(CSRC, "sushi_fft_plan.inc"),
      os.path.join(CSRC, "_gen_dft16_f16_bound.inc"), os.path.join(CSRC, "_gen_dft16_f16_bound_low.inc")]),
]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libsushi_hip.so)")


def write_twiddles(n=16384):
    """exp(-2*pi*i*k/n) as float32 literals (interleaved re, im), correctly rounded from float64."""
    import numpy as np
    k = np.arange(n, dtype=np.float64)
    ang = 2.0 * math.pi * k / n
    tab = np.empty(2 * n, np.float32)
    tab[0::2] = np.cos(ang)
    tab[1::2] = -np.sin(ang)
    text = "".join("%.9ef,%s" % (float(v), "\n" if i % 8 == 7 else " ") for i, v in enumerate(tab))
    if not os.path.exists(TWIDDLE_INC) or open(TWIDDLE_INC).read() != text:
        with open(TWIDDLE_INC, "w") as f:
            f.write(text)
    return TWIDDLE_INC


DFT16_INC = os.path.join(CSRC, "_gen_dft16_f16.inc")


def write_dft16_operands():
    """The B operands of ifft_kernel's first pass on the matrix pipe (fft_core.hpp dft16_operand): for each of the four products
    (real parts of the result: high / low half of the matrix; imaginary parts: high / low) and each lane, eight halves as four
    32-bit words.  Inverse transform (DIR = +1)."""
    import numpy as np
    words = []
    for form in (0, 1):
        vals = np.empty((64, 8), np.float64)
        for l in range(64):
            for j in range(8):
                k, n = 8 * (l >> 4) + j, l & 15
                kk, part = k & 15, k >> 4
                ang = 2.0 * math.pi * ((n * kk) & 15) / 16.0
                wr, wi = math.cos(ang), math.sin(ang)
                vals[l, j] = (wr if part == 0 else -wi) if form == 0 else (wi if part == 0 else wr)
        hi = vals.astype(np.float16)
        lo = (vals - hi.astype(np.float64)).astype(np.float16)
        for part in (hi, lo):
            words.append(np.ascontiguousarray(part).view(np.uint32).reshape(-1))
    flat = np.concatenate(words)
    text = "".join("0x%08xu,%s" % (int(v), "\n" if i % 8 == 7 else " ") for i, v in enumerate(flat))
    if not os.path.exists(DFT16_INC) or open(DFT16_INC).read() != text:
        with open(DFT16_INC, "w") as f:
            f.write(text)
    return DFT16_INC


DFT16H_INC = os.path.join(CSRC, "_gen_dft16_f16_bound.inc")


def write_dft16_bound_operands():
    """The B operands of bound_kernel's first pass (fft_core.hpp fft_wave_half_front): the DFT matrix's high halves only, times
    2^-10 -- [real parts of the result, imaginary parts][lane] x 8 halves as four 32-bit words."""
    import numpy as np
    words = []
    for form in (0, 1):
        vals = np.empty((64, 8), np.float64)
        for l in range(64):
            for j in range(8):
                k, n = 8 * (l >> 4) + j, l & 15
                kk, part = k & 15, k >> 4
                ang = 2.0 * math.pi * ((n * kk) & 15) / 16.0
                wr, wi = math.cos(ang), math.sin(ang)
                vals[l, j] = (wr if part == 0 else -wi) if form == 0 else (wi if part == 0 else wr)
        words.append(np.ascontiguousarray((vals * 2.0 ** -10).astype(np.float16)).view(np.uint32).reshape(-1))
    flat = np.concatenate(words)
    text = "".join("0x%08xu,%s" % (int(v), "\n" if i % 8 == 7 else " ") for i, v in enumerate(flat))
    if not os.path.exists(DFT16H_INC) or open(DFT16H_INC).read() != text:
        with open(DFT16H_INC, "w") as f:
            f.write(text)
    return DFT16H_INC


DFT16L_INC = os.path.join(CSRC, "_gen_dft16_f16_bound_low.inc")


def write_dft16_bound_low_operands():
    """The B operands of bound_low_kernel's first pass (fft_core.hpp fft_wave_half_front_low, dft16_low_operand): the rows of the
    16-point inverse DFT matrix for the eight d1 a low-band group holds, high halves times 2^-10 -- [real parts of the result,
    imaginary parts][lane] x 4 halves as two 32-bit words (K = 16: v_mfma_f32_16x16x16_f16)."""
    import numpy as np
    d1_of = lambda kq, j: kq + (0, 2, 12, 14)[j]
    words = []
    for form in (0, 1):
        vals = np.empty((64, 4), np.float64)
        for l in range(64):
            for j in range(4):
                kq, n = l >> 4, l & 15
                part, d1 = kq >> 1, d1_of(kq & 1, j)
                ang = 2.0 * math.pi * ((n * d1) & 15) / 16.0
                wr, wi = math.cos(ang), math.sin(ang)
                vals[l, j] = (wr if part == 0 else -wi) if form == 0 else (wi if part == 0 else wr)
        words.append(np.ascontiguousarray((vals * 2.0 ** -10).astype(np.float16)).view(np.uint32).reshape(-1))
    flat = np.concatenate(words)
    text = "".join("0x%08xu,%s" % (int(v), "\n" if i % 8 == 7 else " ") for i, v in enumerate(flat))
    if not os.path.exists(DFT16L_INC) or open(DFT16L_INC).read() != text:
        with open(DFT16L_INC, "w") as f:
            f.write(text)
    return DFT16L_INC


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    m = os.path.getmtime(target)
    return any(os.path.getmtime(p) > m for p in deps)


def needs_build():
    if not os.path.exists(TWIDDLE_INC) or not os.path.exists(os.path.join(CSRC, "_gen_dft16_f16.inc")) or \
            not os.path.exists(os.path.join(CSRC, "_gen_dft16_f16_bound.inc")) or not os.path.exists(DFT16L_INC):
        return True
    deps = list(COMMON_DEPS)
    for name, _flags, extra in UNITS:
        deps += [os.path.join(CSRC, name + ".hip")] + extra
    return _stale(LIB, deps)


def build_native(force=False, verbose=False, defines=(), lib=None, obj_tag=""):
    """hipcc --offload-arch=gfx950 -O3 -c csrc/*.hip -> lib/obj/*.o -> lib/libsushi_hip.so
    `defines` / `lib` / `obj_tag`: a variant library beside the product one (tools/ A/B measurements), e.g.
    defines=("-DSUSHI_FFT_LOGN=13",), lib=".../libsushi_hip_n13.so", obj_tag="_n13"."""
    write_twiddles()
    write_dft16_operands()
    write_dft16_bound_operands()
    write_dft16_bound_low_operands()
    lib = lib or LIB
    if not force and not defines and not needs_build():
        return lib
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall"] + list(defines)
    objs, procs = [], []
    for name, flags, extra in UNITS:
        src = os.path.join(CSRC, name + ".hip")
        obj = os.path.join(OBJ_DIR, name + obj_tag + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + COMMON_DEPS + extra):
            cmd = base + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden"] + objs + ["-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
