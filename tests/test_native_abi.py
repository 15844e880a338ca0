"""The C-ABI library: builds, loads and exports every symbol include/sushi_hip.h declares
(no compute calls -- there is no GPU on the CPU test box)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

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
    C = ctypes
    assert L.sushi_hip_abi_version() == _native.ABI_VERSION == 13
    assert L.sushi_hip_strerror(0) == b"ok" and b"invalid" in L.sushi_hip_strerror(-1)
    assert L.sushi_hip_centre(_native.U8) == 128.0 and L.sushi_hip_centre(_native.F32) == 0.5
    N, B = L.sushi_hip_fft_size(), L.sushi_hip_fft_block()
    assert (N, B) == (16384, 4096)
    # sizes are pure host arithmetic
    assert L.sushi_hip_stream_bytes(0, 1, 0) == 0 and L.sushi_hip_stream_bytes(10, 7, 0) == 0
    plain, searchable = L.sushi_hip_stream_bytes(100000, _native.F32, 0), L.sushi_hip_stream_bytes(100000, _native.F32, 1)
    assert plain >= 100000 * 4 + 2 * 100001 * 8 + 100001 * 4 and plain % 256 == 0
    # two blocks + the all-zero one: whole rows, the low-band rows (a quarter of the bins) and three arrays of row norms outside the band
    assert L.sushi_hip_stream_spectra_bytes(4097) == 3 * N * 4 + 3 * N + 3 * 256 and L.sushi_hip_stream_spectra_bytes(0) == 0
    assert searchable - plain == (L.sushi_hip_stream_spectra_bytes(100000) + 255) // 256 * 256
    # argument validation happens before any HIP call
    h = C.c_void_p()
    assert L.sushi_hip_stream_create(None, 1, 10, 0, None, 0, None, C.byref(h)) == -1
    assert L.sushi_hip_stream_create(C.c_void_p(4096), 7, 10, 0, C.c_void_p(4096), 1 << 20, None, C.byref(h)) == -1
    assert L.sushi_hip_stream_create(C.c_void_p(4096), 1, 10, 0, C.c_void_p(4096 + 8), 1 << 20, None, C.byref(h)) == -2
    assert L.sushi_hip_stream_create(C.c_void_p(4096), 1, 10, 0, C.c_void_p(4096), 16, None, C.byref(h)) == -4
    assert L.sushi_hip_batch_run(None, 2e-5, None, None, None) == -1
    assert L.sushi_hip_batch_set_method(None, 0) == -1
    assert L.sushi_hip_batch_create(None, None, None, 0, 0, -1, 0, None, 0, None, C.byref(h)) == -1
    assert L.sushi_hip_load_decode(None, 10, 2, 2, None, None) == -1
    assert L.sushi_hip_load_decode(C.c_void_p(4096), 10, 2, 4, C.c_void_p(4096), None) == -1     # wav.py:75-76 sample widths
    # FFT path geometry: pairs sit on the absolute pair grid (a pair = 2 * (N - B) positions = 6 blocks)
    pairs, segs = C.c_int32(), C.c_int32()
    assert L.sushi_hip_fft_layout(0, 1, 1, C.byref(pairs), C.byref(segs)) == 0
    assert (pairs.value, segs.value) == (1, 1)
    assert L.sushi_hip_fft_layout(6 * B - 1, 2, B + 1, C.byref(pairs), C.byref(segs)) == 0
    assert (pairs.value, segs.value) == (2, 2)                 # positions 6B-1 and 6B lie in different pairs
    assert L.sushi_hip_fft_layout(4095, 4098, 36000, C.byref(pairs), C.byref(segs)) == 0
    assert (pairs.value, segs.value) == (1, 9)
    assert L.sushi_hip_fft_layout(123456, 2880001, 36000, C.byref(pairs), C.byref(segs)) == 0
    assert pairs.value == (123456 + 2880000) // (6 * B) - 123456 // (6 * B) + 1
    assert L.sushi_hip_fft_layout(-1, 1, 1, C.byref(pairs), C.byref(segs)) == -1
    assert _native.fft_layout(4095, 4098, 36000) == (1, 9)
    # the low-band rows: N/4 slots, every bin of the band (f < N/8, f >= 7N/8) exactly once, in the order the bound's transform
    # loads it (fft_core.hpp "LOW BAND": entry = ((g * 4 + g4) * 32 + 16 kq + m'), sub-position j <-> d1 = kq + {0, 2, 12, 14}[j])
    slots = [L.sushi_hip_fft_low_slot_of_bin(f) for f in range(N)]
    band = [f for f in range(N) if f < N // 8 or f >= 7 * N // 8]
    assert sorted(slots[f] for f in band) == list(range(N // 4)) and all(slots[f] == -1 for f in range(N // 8, 7 * N // 8))
    for f in (0, 1, 8, 2047, N - 1, N - 2048, 777, N - 5):
        e, j = divmod(slots[f], 4)
        g, g4, l = e >> 7, (e >> 5) & 3, e & 31
        kq, mm = l >> 4, l & 15
        k = g + 8 * (64 * (kq + (0, 2, 12, 14)[j]) + 4 * (4 * g4 + (mm & 3)) + (mm >> 2))
        assert (k if k < N // 8 else k + N // 2) == f
    # batch sizing: more workspace than one sub-batch needs is not taken; less cuts the batch, never below one request
    req = np.zeros(4, _native.REQUEST_DTYPE)
    req["win_start"] = [100000, 140000, 190000, 300000]
    req["n_pos"], req["tmpl_len"] = 240001, 36000
    req["tmpl_off"] = [0, 40000, 80000, 120000]
    whole = L.sushi_hip_batch_bytes(req.ctypes.data, 4, _native.PATH_FFT, -1, 0)
    assert whole == L.sushi_hip_batch_bytes(req.ctypes.data, 4, _native.PATH_FFT, -1, 1 << 40) and whole > 4 * 10 * N * 8
    small = L.sushi_hip_batch_bytes(req.ctypes.data, 4, _native.PATH_FFT, -1, 1)
    assert 0 < small < whole
    direct = L.sushi_hip_batch_bytes(req.ctypes.data, 4, _native.PATH_DIRECT, -1, 0)
    assert 0 < direct < 1 << 16
    assert L.sushi_hip_batch_bytes(req.ctypes.data, 4, 9, -1, 0) == 0                  # unknown path
    assert L.sushi_hip_batch_bytes(req.ctypes.data, 4, _native.PATH_DIRECT, 99, 0) == 0  # unknown variant
    # lanes: a batch cut into parts that run side by side keeps the one-sub-batch cut beside the parts (the runs that form whole
    # rows throughout take it), so it needs what one sub-batch needs + a second schedule; a cap that cannot hold that falls back
    # to one sub-batch after the other; a large batch takes lanes by itself
    os.environ["SUSHI_HIP_LANES"] = "4:2"
    try:
        lanes = L.sushi_hip_batch_bytes(req.ctypes.data, 4, _native.PATH_FFT, -1, 0)
        assert whole <= lanes < 1.3 * whole, (lanes, whole)                              # (four small searches: a part's fixed buffers dominate)
        assert L.sushi_hip_batch_bytes(req.ctypes.data, 4, _native.PATH_FFT, -1, 1) == small
        os.environ["SUSHI_HIP_LANES"] = "1:1"
        assert L.sushi_hip_batch_bytes(req.ctypes.data, 4, _native.PATH_FFT, -1, 0) == whole
    finally:
        del os.environ["SUSHI_HIP_LANES"]
    big = np.zeros(3000, _native.REQUEST_DTYPE)
    big["win_start"] = np.arange(3000) * 28000
    big["n_pos"], big["tmpl_len"] = 2880001, 36000
    big["tmpl_off"] = np.arange(3000) * 28000
    auto = L.sushi_hip_batch_bytes(big.ctypes.data, 3000, _native.PATH_FFT, -1, 0)
    os.environ["SUSHI_HIP_LANES"] = "1:1"
    try:
        one = L.sushi_hip_batch_bytes(big.ctypes.data, 3000, _native.PATH_FFT, -1, 0)
    finally:
        del os.environ["SUSHI_HIP_LANES"]
    assert one < auto < 1.001 * one, (auto, one)                                        # nine parts on three lanes + the one-sub-batch cut: a second schedule
    req["n_pos"][2] = 0
    assert L.sushi_hip_batch_bytes(req.ctypes.data, 4, _native.PATH_FFT, -1, 0) == 0   # malformed request
    n = C.c_int(-1)
    assert L.sushi_hip_profile_end(None, 0, C.byref(n)) == -1


def test_struct_layouts_match_header(tmp_path):
    src = os.path.join(tmp_path, "layout.c")
    exe = os.path.join(tmp_path, "layout")
    with open(src, "w") as f:
        f.write('#include <stdio.h>\n#include <stddef.h>\n#include "sushi_hip.h"\n'
                'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(SushiHipRequest),'
                'offsetof(SushiHipRequest,tmpl_off),offsetof(SushiHipRequest,win_start),'
                'offsetof(SushiHipRequest,tmpl_len),offsetof(SushiHipRequest,n_pos),'
                'sizeof(SushiHipBatchInfo),sizeof(SushiHipBatchDiag));return 0;}\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, src, "-o", exe])     # header is plain C
    vals = [int(x) for x in subprocess.check_output([exe]).split()]
    d = _native.REQUEST_DTYPE
    assert vals == [d.itemsize, d.fields["tmpl_off"][1], d.fields["win_start"][1], d.fields["tmpl_len"][1],
                    d.fields["n_pos"][1], ctypes.sizeof(_native.BatchInfo), ctypes.sizeof(_native.BatchDiag)]


@pytest.mark.gpu
def test_batch_reset_replans_in_place_and_the_drop_in_call_reuses_its_batch(oracle):
    """sushi_hip_batch_reset: the same handle for other requests -- results are those of a fresh batch; requests that need more
    memory than the batch was created with are refused (ENOSPACE, the batch unchanged); WavStream.find_substream keeps one small
    batch per (source, size, method) and re-plans it call after call."""
    import numpy as np
    from sushi_amd import synth
    from sushi_amd.device import SearchBatch
    from sushi_amd.wav import WavStream
    rate = 12000
    dst_pcm = synth.make_dst_pcm(120, rate, seed=11)
    src_pcm = synth.make_src_pcm(dst_pcm, int(2.5 * rate), seed=12)
    dst = WavStream.from_samples(dst_pcm, rate, sample_type="float32")
    src = WavStream.from_samples(src_pcm, rate, sample_type="float32")
    D, S = dst.device_stream(), src.device_stream()
    reqs = [([200000], [30000], [150000], [200000]), ([400000], [9000], [380000], [120000]), ([600000], [48000], [500000], [300001])]
    b = SearchBatch(D, S, *reqs[0], path="fft", headroom=4.0)
    b.run()
    got = [tuple(x.copy() for x in b.results())]
    for r in reqs[1:]:
        assert b.reset(*r)
        b.run()
        got.append(tuple(x.copy() for x in b.results()))
    for r, (gi, gs) in zip(reqs, got):
        f = SearchBatch(D, S, *r, path="fft")
        f.run()
        fi, fs = f.results()
        assert (fi == gi).all() and (fs.view(np.uint32) == gs.view(np.uint32)).all()
        assert int(gi[0]) + r[2][0] == r[0][0] + int(2.5 * rate)                    # the planted offset
    # more than what a batch was created with: refused, and the batch still answers its own requests
    b2 = SearchBatch(D, S, *reqs[1], path="fft")
    assert not b2.reset(*reqs[2])
    b2.run()
    gi, gs = b2.results()
    assert (gi == got[1][0]).all() and (gs.view(np.uint32) == got[1][1].view(np.uint32)).all()
    with pytest.raises(Exception):
        b.reset([1, 2], [10, 10], [0, 0], [100, 100])                              # a batch keeps its number of searches
    # the drop-in call: one pooled batch per (source stream, size, method), the reference's answers
    odst = oracle.OracleWavStream(dst.data, dst.sample_rate, dst.sample_count, dst.padding_size)
    for s, e, w in ((10.0, 12.5, 8.0), (40.0, 41.0, 20.0), (70.0, 74.0, 1.5), (20.0, 23.0, 30.0)):
        p = src.get_substream(s, e)
        d, t = dst.find_substream(p, s + 2.5, w)
        rd, rt = odst.find_substream(p, s + 2.5, w)
        assert abs(t - rt) <= 1e-12 and abs(float(d) - float(rd)) <= 1e-4 * float(rd) + 2.5e-7
    assert len(dst._small_batches) == 1
