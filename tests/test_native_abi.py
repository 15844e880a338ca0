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
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", path]).decode()
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and "sushi_hip_" in l)
    assert exported == declared                      # nothing undeclared leaks out either


def test_host_only_entry_points():
    L = _native.lib()
    assert L.sushi_hip_abi_version() == 5
    assert L.sushi_hip_strerror(0) == b"ok" and b"invalid" in L.sushi_hip_strerror(-1)
    assert _native.variant_tiles() == [1024, 4096, 16384]
    assert L.sushi_hip_variant_tile_positions(99) == -1
    assert L.sushi_hip_centre(_native.U8) == 128.0 and L.sushi_hip_centre(_native.F32) == 0.5
    assert L.sushi_hip_prepare_base_bytes(4096) == 2 * 2 * 8          # nb = 1: bases of block 0 and of sample n, two sums
    assert L.sushi_hip_prepare_base_bytes(4097) == 2 * 3 * 8
    # argument validation happens before any HIP call
    assert L.sushi_hip_prepare_stream(None, 1, 10, None, None, None, None, None, 0, None) == -1
    assert L.sushi_hip_match_batch(None, None, None, 0, None, None, None, 0, 0.5, 0, None, 0, 0, 0,
                                   None, None, None, None) == -1
    # FFT path: layout helpers are pure host arithmetic
    import ctypes as C
    assert L.sushi_hip_fft_hop() == 4096
    assert L.sushi_hip_spectra_blocks(1) == 1 and L.sushi_hip_spectra_blocks(4096) == 1
    assert L.sushi_hip_spectra_blocks(4097) == 2 and L.sushi_hip_spectra_blocks(0) == 0
    assert L.sushi_hip_spectra_bytes(4097) == 3 * 8192 * 8 and L.sushi_hip_spectra_bytes(0) == 0
    pairs, segs = C.c_int32(), C.c_int32()
    assert L.sushi_hip_fft_layout(0, 1, 1, C.byref(pairs), C.byref(segs)) == 0
    assert (pairs.value, segs.value) == (1, 1)
    assert L.sushi_hip_fft_layout(4095, 2, 4097, C.byref(pairs), C.byref(segs)) == 0
    assert (pairs.value, segs.value) == (1, 2)                 # positions 4095, 4096 -> blocks 0 and 1 -> one pair
    assert L.sushi_hip_fft_layout(4095, 4098, 36000, C.byref(pairs), C.byref(segs)) == 0
    assert (pairs.value, segs.value) == (2, 9)                 # blocks 0..2 -> two pairs
    assert L.sushi_hip_fft_layout(-1, 1, 1, C.byref(pairs), C.byref(segs)) == -1
    for w, p, m in [(0, 1, 1), (4095, 2, 4097), (123456, 1440001, 36000), (8192, 4096, 65537)]:
        assert L.sushi_hip_fft_layout(w, p, m, C.byref(pairs), C.byref(segs)) == 0
        vp, vs = _native.fft_layout([w], [p], [m])
        assert (int(vp[0]), int(vs[0])) == (pairs.value, segs.value)
    assert L.sushi_hip_fft_workspace_bytes(1, 1, 1) == 65536 + 65536 + 3 * 256
    assert L.sushi_hip_fft_workspace_bytes(176555, 9379, 1000) > L.sushi_hip_fft_workspace_bytes(176555, 9379, 1)
    assert L.sushi_hip_prepare_spectra(None, 1, 10, None, 0, None) == -1
    assert L.sushi_hip_match_batch_fft(None, None, None, 0, None, None, None, None, None, None, 0, None, None, 1, 0, None, None,
                                       0, 2e-5, None, 0, None, None, None, None, None, None) == -1
    # the inverse-transform schedule is a permutation of each sub-batch's pairs
    win = np.array([100000, 140000, 190000, 300000], np.int64)       # equal shapes: with the smallest workspace
    npos = np.array([240001, 240001, 240001, 240001], np.int64)       # no two searches share a sub-batch
    mlen = np.array([36000, 36000, 36000, 36000], np.int64)
    vp_, vs_ = _native.fft_layout(win, npos, mlen)
    desc = np.zeros(4, _native.SEARCH_DTYPE)
    desc["win_start"], desc["n_pos"], desc["tmpl_len"] = win, npos, mlen
    desc["first_pair"][1:] = np.cumsum(vp_[:-1]); desc["first_seg"][1:] = np.cumsum(vs_[:-1])
    total = int(vp_.sum())
    for ws in (1 << 40, int(L.sushi_hip_fft_workspace_bytes(int(vp_.max()), int(vs_.max()), 1))):
        order = np.full(total, -1, np.int32)
        assert L.sushi_hip_fft_pair_order(desc.ctypes.data, 4, ws, order.ctypes.data, total) == 0
        if ws > (1 << 39):
            assert sorted(order.tolist()) == list(range(total))          # one sub-batch
        else:                                                             # one search per sub-batch
            lo = 0
            for k in range(4):
                assert sorted(order[lo:lo + int(vp_[k])].tolist()) == list(range(int(vp_[k])))
                lo += int(vp_[k])
    assert L.sushi_hip_fft_pair_order(desc.ctypes.data, 4, 1 << 40, order.ctypes.data, total + 1) == -1
    n = C.c_int(-1)
    assert L.sushi_hip_profile_end(None, 0, C.byref(n)) == -1


def test_descriptor_layout_matches_header(tmp_path):
    src = os.path.join(tmp_path, "layout.c")
    exe = os.path.join(tmp_path, "layout")
    with open(src, "w") as f:
        f.write('#include <stdio.h>\n#include <stddef.h>\n#include "sushi_hip.h"\n'
                'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(SushiHipSearch),'
                'offsetof(SushiHipSearch,tmpl_off),offsetof(SushiHipSearch,win_start),'
                'offsetof(SushiHipSearch,tmpl_len),offsetof(SushiHipSearch,n_pos),'
                'offsetof(SushiHipSearch,first_tile),offsetof(SushiHipSearch,first_pair),'
                'offsetof(SushiHipSearch,first_seg));return 0;}\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, src, "-o", exe])     # header is plain C
    vals = [int(x) for x in subprocess.check_output([exe]).split()]
    d = _native.SEARCH_DTYPE
    assert vals == [d.itemsize, d.fields["tmpl_off"][1], d.fields["win_start"][1], d.fields["tmpl_len"][1],
                    d.fields["n_pos"][1], d.fields["first_tile"][1], d.fields["first_pair"][1],
                    d.fields["first_seg"][1]]
