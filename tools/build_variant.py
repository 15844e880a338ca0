#!/usr/bin/env python3
"""Build a VARIANT of libsushi_hip.so next to the product library, for A/B runs on one box (tools/gpu_r3_m.sh:
VARIANTS="name ..." -> sushi_amd/lib/libsushi_hip_<name>.so, picked by SUSHI_HIP_LIB).  The product sources carry no
development switches: a variant is the tree of a commit (default HEAD, i.e. WITHOUT uncommitted changes) plus patches and / or
sed-style substitutions, built in a scratch copy.

usage: build_variant.py NAME [--rev REV] [--patch FILE ...] [--sub 'FILE:OLD:NEW' ...] [--worktree]
  --rev REV      the commit the variant starts from (default HEAD); `--worktree` takes the working tree as it is instead
  --patch FILE   applied in order with `git apply` (tools/experiments/*.patch)
  --sub F:O:N    replace the literal text O by N in file F (must occur exactly once), e.g.
                 --sub 'sushi_amd/csrc/sushi_fft.hip:constexpr int GQ = 4;:constexpr int GQ = 6;'
examples:
  tools/build_variant.py prev --rev HEAD~1
  tools/build_variant.py mfma --patch tools/experiments/r04_ifft_mfma_first_pass.patch
  tools/build_variant.py mfma2 --patch tools/experiments/r04_ifft_mfma_first_pass.patch --patch tools/experiments/r04_ifft_mfma_plan_v2_on_top.patch
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--rev", default="HEAD")
    ap.add_argument("--patch", action="append", default=[])
    ap.add_argument("--sub", action="append", default=[])
    ap.add_argument("--worktree", action="store_true")
    ap.add_argument("--keep", action="store_true", help="keep the scratch copy (its path is printed)")
    args = ap.parse_args()
    scratch = tempfile.mkdtemp(prefix="sushi_variant_%s_" % args.name)
    try:
        if args.worktree:
            for d in ("sushi_amd", "include"):
                shutil.copytree(os.path.join(ROOT, d), os.path.join(scratch, d),
                                ignore=shutil.ignore_patterns("lib", "__pycache__", "_gen_*.inc"))
        else:
            tar = subprocess.Popen(["git", "-C", ROOT, "archive", args.rev, "sushi_amd", "include", "tests"], stdout=subprocess.PIPE)
            subprocess.check_call(["tar", "-x", "-C", scratch], stdin=tar.stdout)
            if tar.wait() != 0:
                raise SystemExit("git archive %s failed" % args.rev)
        for p in args.patch:
            subprocess.check_call(["git", "apply", "--unsafe-paths", "--directory", scratch, os.path.abspath(p)], cwd=scratch)
        for spec in args.sub:
            f, old, new = spec.split(":", 2)
            path = os.path.join(scratch, f)
            text = open(path).read()
            if text.count(old) != 1:
                raise SystemExit("--sub: %r occurs %d times in %s (want exactly 1)" % (old, text.count(old), f))
            open(path, "w").write(text.replace(old, new))
        subprocess.check_call([sys.executable, "-c", "from sushi_amd import build; print(build.build_native(force=True))"], cwd=scratch)
        out = os.path.join(ROOT, "sushi_amd", "lib", "libsushi_hip_%s.so" % args.name)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        shutil.copy2(os.path.join(scratch, "sushi_amd", "lib", "libsushi_hip.so"), out)
        print(out)
    finally:
        if args.keep:
            print("scratch copy kept:", scratch)
        else:
            shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
