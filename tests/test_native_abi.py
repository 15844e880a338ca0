"""The C-ABI library: builds, loads and exports every symbol include/sushi_hip.h declares
(no compute calls -- there is no GPU on the CPU test box)."""
import ctypes
import os
import subprocess

import numpy as np

from sushi_amd import _native, build


def test_library_builds_and_exports_all_declared_symbols():
    path = build.build_native()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    declared = _native.declared_symbols()
    assert len(declared) >= 9
    for name in declared:
        assert hasattr(lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", path]).decode()
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and "sushi_hip_" in l)
    assert exported == declared                      # nothing undeclared leaks out either


def test_host_only_entry_points():
    L = _native.lib()
    assert L.sushi_hip_abi_version() == 1
    assert L.sushi_hip_strerror(0) == b"ok" and b"invalid" in L.sushi_hip_strerror(-1)
    assert _native.variant_tiles() == [1024, 4096, 16384]
    assert L.sushi_hip_variant_tile_positions(99) == -1
    assert L.sushi_hip_centre(_native.U8) == 128.0 and L.sushi_hip_centre(_native.F32) == 0.5
    assert L.sushi_hip_prepare_workspace_bytes(4096) == 16
    assert L.sushi_hip_prepare_workspace_bytes(4097) == 32
    # argument validation happens before any HIP call
    assert L.sushi_hip_prepare_stream(None, 1, 10, None, None, None, None, 0, None) == -1
    assert L.sushi_hip_match_batch(None, None, None, 0, None, None, None, 0, 0.5, 0, None, 0, 0, 0,
                                   None, None, None, None) == -1


def test_descriptor_layout_matches_header(tmp_path):
    src = os.path.join(tmp_path, "layout.c")
    exe = os.path.join(tmp_path, "layout")
    with open(src, "w") as f:
        f.write('#include <stdio.h>\n#include <stddef.h>\n#include "sushi_hip.h"\n'
                'int main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(SushiHipSearch),'
                'offsetof(SushiHipSearch,tmpl_off),offsetof(SushiHipSearch,win_start),'
                'offsetof(SushiHipSearch,tmpl_len),offsetof(SushiHipSearch,n_pos),'
                'offsetof(SushiHipSearch,first_tile));return 0;}\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, src, "-o", exe])     # header is plain C
    vals = [int(x) for x in subprocess.check_output([exe]).split()]
    d = _native.SEARCH_DTYPE
    assert vals == [d.itemsize, d.fields["tmpl_off"][1], d.fields["win_start"][1], d.fields["tmpl_len"][1],
                    d.fields["n_pos"][1], d.fields["first_tile"][1]]
