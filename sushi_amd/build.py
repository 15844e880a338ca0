"""Build libsushi_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "sushi_hip.hip")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "sushi_hip.h")
LIB_DIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIB_DIR, "libsushi_hip.so")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libsushi_hip.so)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    m = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > m for p in (SRC, HEADER))


def build_native(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -O3 -shared -fPIC csrc/sushi_hip.hip -> lib/libsushi_hip.so
    -ffp-contract=off: the float64 epilogue restates cv2's operation order; a fused a*b-c*d would
    round differently from the reference (the hot loop is MFMA builtins, unaffected)."""
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fvisibility=hidden", "-ffp-contract=off", "-Wall", SRC, "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
