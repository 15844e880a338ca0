// sushi_amd/csrc/sushi_fft.hip -- overlap-save FFT form of Sushi's template match for gfx950 (MI355X),
// and the batch handle of the C ABI.
//
// For every (pattern, window) request the float32 result.argmin and result[argmin] of
//   wav.py:185  result = cv2.matchTemplate(search_source, pattern, cv2.TM_SQDIFF_NORMED)
//   wav.py:186  min_idx = result.argmin(axis=1)[0]
// cv2's crossCorr() computes the sliding dot product by block DFT; so does this path, which turns
// the O(P*M) FLOP-bound search into an O(P log N) HBM-bound one (DESIGN.md "FFT path").  With N-point
// complex transforms, pattern segments of B = FFT_SEG samples and H = N - B valid positions per real block:
//
//   spectra_kernel   once per destination stream, for every block j (hop B):
//                    Z_j = DFT_N( x[jB .. jB+N) + i * x[jB+H .. jB+H+N) )
//                    (+ the low band of Z_j once more and its norms outside the band: the band-split exclusion below)
//   tspec_kernel     per search, per pattern segment s (B samples, zero padded to N):
//                    Tt_s = conj(DFT_N(t_s)) / N                                  (+ its low band and its norm outside the band)
//   mac_kernel       per search, per frequency bin f, per pair I of the absolute pair grid (pair I starts at
//                    block FFT_STEP * I, FFT_STEP = 2H / B):
//                    Y_I(f) = sum_s Tt_s(f) * Z_{FFT_STEP*I+s}(f)       (a 1-D Toeplitz product along j)
//                    -- over every pair on the LOW rows (a quarter of the bins: the band-split form of the exclusion), or on whole
//                    rows (the whole-row form, no exclusion, the dense fall-back); mac_list_kernel / mac_rows_kernel form the
//                    whole rows of LISTED pairs only
//   bound_low_kernel / bound_kernel + slb_kernel, pilot_kernel, survivor_kernel
//                    a LOWER bound of every pair's scores without scoring it; the pair with the smallest bound of every search
//                    is transformed first, then only the pairs whose bound is not above what the search has found
//   ifft_kernel      y_I = IDFT_N(Y_I): Re y_I[r] / Im y_I[r] (r < H) are the cross terms of positions
//                    FFT_STEP*I*B + r and FFT_STEP*I*B + H + r.  Fused epilogue: window energies, normalised f32
//                    score, the pair's error bound, arg-min, and the list of positions that can still be the
//                    minimum (candidates); the pair's lower bound held to what it really scores (the exclusion's audit).
//   refine_kernel    (sushi_hip.hip) exact float64 re-evaluation of the candidates -> final (index, score);
//   collect + tiles  searches with more candidates than the lists hold: the inverse transforms of their pairs
//                    are redone with the search's final threshold, every candidate goes to a per-tile list
//                    (a tile = 1024 positions) and exact_tiles_kernel (sushi_hip.hip) evaluates those exactly.
//
// Pairing two real blocks as one complex block makes every N-point complex DFT produce 2H useful
// results and needs no real-FFT untangling pass: the pattern is real, so correlation is linear
// over the real and imaginary parts.
//
// gfx950 only: wave64, 160 KiB LDS/CU.  No CUDA compatibility paths.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <new>
#include <type_traits>
#include <vector>

#include "../../include/sushi_hip.h"
#include "sushi_common.hpp"
#include "sushi_internal.hpp"
#include "fft_core.hpp"
#include "mac_core.hpp"

namespace {

using namespace sushi;
using sushi_fft::cpx;

constexpr int FN = FFT_N;                              // complex points per transform
constexpr int FT = sushi_fft::Plan<FFT_LOGN>::NT;      // threads per transform workgroup
constexpr int FH = FFT_H;                              // result positions per half of a pair
constexpr int HPT = FH / FT;                           // result positions per thread and half
constexpr int RPB = FFT_SEG / FT;                      // of which per block
static_assert(FH % FT == 0 && FFT_SEG % FT == 0 && HPT == FFT_VB * RPB, "position layout");
static_assert(HPT <= sushi_fft::PER, "a thread's valid outputs are a prefix of its transform outputs");
constexpr int LDS_FLOATS = sushi_fft::lds_floats<FFT_LOGN>();

// ---- block spectra, pattern spectra and their products Y are all kept as packed halves -----------------------------
// (half the bytes of everything the step moves between kernels, and two multiply-adds per v_dot2_f32_f16 in mac_kernel)
// Two things make 11 bits enough for a stage that only ranks:
//  * block spectra are of the CENTRED destination samples (x - c), patterns stay as they are:
//        y'[p] = sum_m T[m] (I[p+m] - c) = sum T I - c sum T
//    -- the correction is one constant per search -- and Y then carries no product of two DC terms: its energy, and with
//    it the quantisation noise of every position, is that of pattern x centred audio instead of ~M/4 at every position;
//  * the noise is modelled per pair from the energy of the Y row actually loaded (Parseval), added to the pair's error
//    bound, and checked like the rest of the bound (candidates and one audited non-candidate per search, refine_kernel).
//    Rounding the two factors to halves before they are multiplied perturbs a product Tt_s(f) Z_j(f) by the same relative
//    2^-11 per factor as rounding the sum does afterwards: with the terms of a bin's sum taken as independent that is two
//    more times the variance of the stored row's own rounding (pair_error_model).
// The constant c is the stream's own mean (any constant is exact; the mean keeps DC out whatever level the data sits at).
// Power-of-two scales keep every stored half inside the format whatever the magnitude of the data, and away from its
// subnormals: block spectra by the stream (|Z_j(f)| <= sqrt(2 N E7), E7 = the largest centred energy of FFT_STEP + 1
// consecutive blocks, SushiHipStream.stats), pattern spectra by the pattern (|Tt_s(f)| <= 64 |T| / N), and the float32
// sums of their products are brought to the scale of Y when they are stored:
//   |Y(f)| <= sum_s |Tt_s(f)| * max_j |Z_j(f)| <= (64 sqrt(n_seg) |T| / N) * (sqrt(7 * 4096) sqrt(E7));
// typical values sit ~sqrt(N) below these bounds, twenty binary orders above the smallest normal half.
__device__ __forceinline__ float pow2_under(double target, double bound) {
    if (!(bound > 0.0)) return 1.0f;
    int k = (int)floor(log2(target / bound));
    k = k < -60 ? -60 : (k > 60 ? 60 : k);
    return (float)ldexp(1.0, k);
}
__device__ __forceinline__ float y_scale_for(double tnorm, int n_seg, double e7) {
    return pow2_under(32768.0, (64.0 * sqrt((double)n_seg) * tnorm / (double)FN) * (169.33 * sqrt(e7)));
}
__device__ __forceinline__ float z_scale_for(double e7) { return pow2_under(32768.0, 181.02 * sqrt(e7)); }
__device__ __forceinline__ float t_scale_for(double tnorm) { return pow2_under(8192.0, 64.0 * tnorm / (double)FN); }
// one complex number as a packed half pair (re | im << 16), round to nearest even, never infinite
__device__ __forceinline__ unsigned pack_h2(float re, float im) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 q = {(_Float16)__builtin_amdgcn_fmed3f(re, -65504.f, 65504.f), (_Float16)__builtin_amdgcn_fmed3f(im, -65504.f, 65504.f)};
    return __builtin_bit_cast(unsigned, q);
}
// sum of |re|^2 + |im|^2 over the four stored words (re | im << 16) of one 16-byte entry, added to `acc`.
// The entry is cast to EIGHT halves as a whole and taken apart by sub-vectors: hipcc 7.2 compiles the obvious form --
// `bit_cast<half2>(entry[j])` for j = 0 .. 3 in an unrolled loop -- to four reads of the entry's FIRST word (it narrows the 16-byte
// load to a dword: `v_dot2c_f32_f16 v7, v2, v2` four times over; a ten-line reproducer is in tools/experiments/README.md).  The row
// energies behind the error model were therefore four times every fourth bin: right on noise-like rows, a factor 10^4 short on
// a row whose energy sits in one bin.  Found in round 6 with bursts of a tone (tests/test_bound_stress.py "fs8burst").
__device__ __forceinline__ float add_abs2_entry(const sushi_fft::uint4v e, float acc) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const h8 v = __builtin_bit_cast(h8, e);
    acc = __builtin_amdgcn_fdot2(v.s01, v.s01, acc, false);
    acc = __builtin_amdgcn_fdot2(v.s23, v.s23, acc, false);
    acc = __builtin_amdgcn_fdot2(v.s45, v.s45, acc, false);
    return __builtin_amdgcn_fdot2(v.s67, v.s67, acc, false);
}
constexpr int ROW_BYTES = FN * 4;              // a stored spectrum: one 32-bit word per bin
constexpr int ROWE = FN / sushi_mac::BINS;     // ... as 16-byte entries (four bins: what a lane of mac_kernel owns)
constexpr float Y_KQ = 8.0f;                  // the quantisation term of a pair's bound, in standard deviations
// The low band of every spectrum (bins |f| < N/8) is kept a second time, as rows of LROWE entries in the order bound_low_kernel
// loads them (fft_core.hpp "LOW BAND"): the band-split exclusion multiplies, stores and transforms only these.
constexpr int LROWE = sushi_fft::LB_ENTRIES;   // 16-byte entries of a low row
constexpr int LROW_BYTES = LROWE * 16;
static_assert(FFT_LOGN == 14 && sushi_fft::W_LDS_FLOATS <= LDS_FLOATS && FT == sushi_fft::WNT, "the wave plan is the 16384-point inverse");

// Spectra are STORED in the order the inverse transform loads them (fft_core.hpp "Wave plan": wslot_of_bin): block
// spectra, pattern spectra and their products only have to agree on one order of the bins.  A forward transform ends with
// thread tid holding X[tid + 1024 r] in register r.
// That is a permutation of the THREADS (same register index): the bins register r of thread (w, l) loads are held, after
// a forward transform, by register r of thread w + 64 (l & 15) + 16 (l >> 4).  The forward kernels hand their outputs
// over through the LDS (real parts, then imaginary parts; position tid + tid / 64 makes the gather conflict-free) so that
// the global stores are whole KiB per wave instead of 16-byte pieces 8 KiB apart (pattern spectra are written every run).
__device__ __forceinline__ void to_load_order(cpx (&v)[sushi_fft::PER], const int tid, float* lds) {
    // element e of the stored order sits at e + e / 16 + e / 1024 (the producer's scattered stores then spread over the banks);
    // a thread takes its four runs of four (the entries it will store) back out
    auto pos = [](const int e) { return e + (e >> 4) + (e >> 10); };
    __syncthreads();                                             // the transform's own use of the buffer is over
#pragma unroll
    for (int r = 0; r < sushi_fft::PER; ++r) lds[pos(sushi_fft::mslot_of_bin(tid + FT * r))] = v[r].x;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < sushi_fft::PER; ++r) v[r].x = lds[pos(4 * sushi_fft::wslot_uint4(tid, r >> 2) + (r & 3))];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < sushi_fft::PER; ++r) lds[pos(sushi_fft::mslot_of_bin(tid + FT * r))] = v[r].y;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < sushi_fft::PER; ++r) v[r].y = lds[pos(4 * sushi_fft::wslot_uint4(tid, r >> 2) + (r & 3))];
}
static_assert(FN + FN / 16 + FN / 1024 <= LDS_FLOATS, "the hand-over fits the transform's buffer");

// exp(-2*pi*i*n/16384), n = 0..16383, float32 rounded from float64 (generated by sushi_amd/build.py)
__device__ const float g_twiddle[2 * sushi_fft::TWIDDLE_N] = {
#include "_gen_twiddle16384.inc"
};

__device__ __forceinline__ const cpx* twiddles() { return reinterpret_cast<const cpx*>(g_twiddle); }

// B operands of the inverse transform's first pass on the matrix pipe: [product][lane] x 8 halves (generated by sushi_amd/build.py)
__device__ __attribute__((aligned(16))) const unsigned g_dft16_b[4 * 64 * 4] = {
#include "_gen_dft16_f16.inc"
};
__device__ __forceinline__ sushi_fft::MfmaB dft16_operands(const int tid) {
    return sushi_fft::load_mfma_b(tid, reinterpret_cast<const sushi_fft::uint4v*>(g_dft16_b));
}
// ... and of bound_kernel's first pass: the matrix's high halves times 2^-10 (generated by sushi_amd/build.py)
__device__ __attribute__((aligned(16))) const unsigned g_dft16_bh[2 * 64 * 4] = {
#include "_gen_dft16_f16_bound.inc"
};

// ... and of bound_low_kernel's (K = 16: the eight d1 a low-band group holds; generated by sushi_amd/build.py)
__device__ __attribute__((aligned(16))) const unsigned g_dft16_bl[2 * 64 * 2] = {
#include "_gen_dft16_f16_bound_low.inc"
};

// What a forward transform leaves for the band-split exclusion.  Thread tid ends with X[tid + 1024 r] in register r: the low band
// is r = 0, 1, 14, 15 of every thread -- one 16-byte entry of the low row (fft_core.hpp lslot_of_thread) --, and of the other
// twelve bins the energy of the halves AS STORED is summed: |sum over the bins outside the band of Tt_s(f) Z_j(f)| is at most
// the product of the two rows' norms outside the band (Cauchy-Schwarz), whatever the phases.
// The band is kept MIRROR-SYMMETRIC: bin 7N/8 (thread 0's register 14) has its mirror N/8 outside the band, so it is counted with
// the rest -- its low-row entry is zero and its energy goes to the norm -- and the band is |f| < N/8 strictly.  slb_kernel's split
// of the rest into the two real blocks' parts (real_block_rest_norms) holds over a mirror-symmetric set of bins only.
__device__ __forceinline__ uint4 low_entry_and_rest(const cpx (&v)[sushi_fft::PER], const float sc, const int tid, float& rest2) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    float e = 0.f;
#pragma unroll
    for (int r = 2; r < 14; ++r) {
        const h2 h = __builtin_bit_cast(h2, pack_h2(v[r].x * sc, v[r].y * sc));
        e = __builtin_amdgcn_fdot2(h, h, e, false);
    }
    unsigned e14 = pack_h2(v[14].x * sc, v[14].y * sc);
    if (tid == 0) {
        const h2 h = __builtin_bit_cast(h2, e14);
        e = __builtin_amdgcn_fdot2(h, h, e, false);
        e14 = 0u;
    }
    rest2 = e;
    return uint4{pack_h2(v[0].x * sc, v[0].y * sc), pack_h2(v[1].x * sc, v[1].y * sc), e14, pack_h2(v[15].x * sc, v[15].y * sc)};
}
// the low entries of a workgroup into their row (whole KiB per wave through the LDS) and the norm of the rest (red: FT / 64 floats)
__device__ __forceinline__ float wave_sum_shfl(float w) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) w += __shfl_xor(w, d, 64);
    return w;
}
__device__ __forceinline__ void store_low_row(const uint4 low, const float rest2, const int tid, float* lds, float* red,
                                              uint4* __restrict__ low_row, float* __restrict__ norm_out) {
    const float w = wave_sum_shfl(rest2);
    __syncthreads();                                             // the hand-over's use of the buffer is over
    uint4* l4 = reinterpret_cast<uint4*>(lds);
    // (one entry of padding per 128: the eight groups a wave's lanes scatter to are then eight different banks)
    const int pos = sushi_fft::lslot_of_thread(tid);
    l4[pos + (pos >> 7)] = low;
    if ((tid & 63) == 0) red[tid >> 6] = w;
    __syncthreads();
    low_row[tid] = l4[tid + (tid >> 7)];
    if (tid == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < FT / 64; ++i) t += red[i];
        *norm_out = sqrtf(t) * 1.000002f;
    }
}
// A block spectrum packs TWO real blocks, Z = A + i B (A, B the conjugate-symmetric spectra of the real blocks at j B and
// j B + H): the real parts of a pair's transform outputs come from A alone, the imaginary parts from B alone, and
// |A|^2 + |B|^2 = |Z|^2 over a symmetric set of bins -- so bounding the two parts separately, each from its own block's norm,
// saves the factor sqrt(2) a bound from |Z| pays.  A(f) = (Z(f) + conj Z(N - f)) / 2, B(f) = (Z(f) - conj Z(N - f)) / 2i, of the
// halves AS STORED; bin N - f of thread tid's register r is register 15 - r of thread FT - tid (tid > 0; tid 0: register
// (16 - r) % 16 of itself).  out[0 / 1] = the norms of A / B over the bins outside the band.
__device__ __forceinline__ void real_block_rest_norms(const cpx (&v)[sushi_fft::PER], const float sc, const int tid, float* lds,
                                                      float* red, float* __restrict__ out_a, float* __restrict__ out_b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    unsigned* w = reinterpret_cast<unsigned*>(lds);
    __syncthreads();
#pragma unroll
    for (int r = 1; r < 15; ++r) w[(r - 1) * FT + tid] = pack_h2(v[r].x * sc, v[r].y * sc);   // (registers 1 and 14: tid 0's partners of 15 and 2 are not needed; kept simple)
    __syncthreads();
    const int pt = tid == 0 ? 0 : FT - tid;
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int r = 2; r < 14; ++r) {
        const int pr = tid == 0 ? 16 - r : 15 - r;                  // 2 .. 13 -> 13 .. 2 (tid > 0), 14 .. 3 (tid 0)
        const h2 z = __builtin_bit_cast(h2, pack_h2(v[r].x * sc, v[r].y * sc));
        const h2 m = __builtin_bit_cast(h2, w[(pr - 1) * FT + pt]);
        const float ar = (float)z.x + (float)m.x, ai = (float)z.y - (float)m.y;      // Z(f) + conj Z(N - f)
        const float br = (float)z.x - (float)m.x, bi = (float)z.y + (float)m.y;      // Z(f) - conj Z(N - f)
        // (bin 7N/8 -- thread 0's register 14, counted with the rest: low_entry_and_rest -- is the mirror of bin N/8, thread 0's
        // register 2: the same moduli once more)
        const float twice = tid == 0 && r == 2 ? 2.f : 1.f;
        sa += twice * (ar * ar + ai * ai);
        sb += twice * (br * br + bi * bi);
    }
    sa = wave_sum_shfl(sa); sb = wave_sum_shfl(sb);
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = sa; red[FT / 64 + (tid >> 6)] = sb; }
    __syncthreads();
    if (tid == 0) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int i = 0; i < FT / 64; ++i) { ta += red[i]; tb += red[FT / 64 + i]; }
        *out_a = sqrtf(0.25f * ta) * 1.000004f;
        *out_b = sqrtf(0.25f * tb) * 1.000004f;
    }
}
static_assert(LROWE == FT && LROW_BYTES + LROWE / 128 * 16 <= LDS_FLOATS * 4 && 14 * FT <= LDS_FLOATS, "one low entry per forward thread");

// ------------------------------------------------------------------------------------------
// Destination-stream spectra
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(FT)
void spectra_kernel(const T* __restrict__ raw, int64_t n, uint32_t* __restrict__ spec, const double* __restrict__ stats,
                    uint4* __restrict__ spec_low, float* __restrict__ znorm_rest, const int64_t norm_stride) {
    const float centre = (float)stats[1];
    const float sz = z_scale_for(stats[0]);
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    __shared__ float red[2 * (FT / 64)];
    const int tid = threadIdx.x;
    const sushi_fft::Twiddles tw = sushi_fft::load_twiddles<FFT_LOGN, -1>(tid, twiddles());
    const int64_t j = blockIdx.x;
    const int64_t base = j * FFT_SEG;
    cpx v[sushi_fft::PER];
#pragma unroll
    for (int r = 0; r < sushi_fft::PER; ++r) {
        // unconditional loads from clamped addresses + select: the loads stay batched
        const int64_t e = base + sushi_fft::in_index<FFT_LOGN>(tid, r);
        const float xa = (float)raw[e < n ? e : n - 1] - centre;      // centred (above); zeros past the end of the stream
        const float xb = (float)raw[(e + FH) < n ? (e + FH) : n - 1] - centre;
        v[r].x = e < n ? xa : 0.f;
        v[r].y = (e + FH) < n ? xb : 0.f;
    }
    sushi_fft::fft_split<FFT_LOGN, -1>(v, tid, lds, tw);
    float rest2;
    const uint4 low = low_entry_and_rest(v, sz, tid, rest2);
    real_block_rest_norms(v, sz, tid, lds, red, znorm_rest + norm_stride + j, znorm_rest + 2 * norm_stride + j);
    to_load_order(v, tid, lds);
    uint4* __restrict__ out = reinterpret_cast<uint4*>(spec + (size_t)j * FN);
#pragma unroll
    for (int u = 0; u < sushi_fft::PER / 4; ++u)
        out[sushi_fft::wslot_uint4(tid, u)] = uint4{pack_h2(v[4 * u].x * sz, v[4 * u].y * sz), pack_h2(v[4 * u + 1].x * sz, v[4 * u + 1].y * sz),
                                                    pack_h2(v[4 * u + 2].x * sz, v[4 * u + 2].y * sz), pack_h2(v[4 * u + 3].x * sz, v[4 * u + 3].y * sz)};
    store_low_row(low, rest2, tid, lds, red, spec_low + (size_t)j * LROWE, znorm_rest + j);
}

// last search of [0, n) whose first_seg is <= x
__device__ __forceinline__ int find_search_by_seg(const SearchDesc* __restrict__ s, int n, int x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (s[mid].first_seg <= x) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// per-search constants of the f32 scoring epilogue, computed once (float64) by tspec_kernel
struct TemplConsts {
    double tU;           // sum T^2 (uncentred)
    float inv_tnorm;     // 1 / sqrt(sum T^2)
    float tnorm;         // sqrt(sum T^2)
    // TM_CCOEFF_NORMED (cv2's numType == 1 statistics, sushi_common.hpp templ_stats)
    float tmean;         // mean T
    float inv_tnorm_c;   // 1 / sqrt(sum (T - mean T)^2); 0 for a flat pattern
    float inv_m;         // 1 / M
    int flat;            // the pattern has no variance: cv2's result is all ones
    float c_sum_t;       // c * sum T: sum T I = y' + c_sum_t (block spectra are of the centred destination samples)
    float inv_scale;     // 1 / the power-of-two scale of this search's stored products Y
    float mac_scale;     // what mac_kernel multiplies its float32 sums by when it stores them: scale of Y / (scale of Tt * scale of Z)
};

// ------------------------------------------------------------------------------------------
// Pattern-segment spectra: Tt = conj(DFT(t_s zero padded)) / N, stored as the packed halves (Re Tt, -Im Tt) mac_kernel's
// dot products take (mac_core.hpp) -- the scaled forward transform itself.  The workgroup of a search's first
// segment also writes the search's scoring constants and the pair -> search map ifft_kernel reads.
// ------------------------------------------------------------------------------------------
struct TspecArgs {
    const void* src_raw;              // the source stream's samples as they are (uint8 or float32)
    const SearchDesc* searches;       // the sub-batch's searches
    int n_sub;
    int sub_first_seg;
    int sub_first_pair;
    uint32_t* tspec;                  // [segments of the sub-batch][FN] packed halves
    int* pairmap;                     // [pairs of the sub-batch] -> search index inside the sub-batch
    struct TemplConsts* tconst;       // [searches of the sub-batch]
    const double* src_s1;
    const double* src_s2;
    double centre;
    const double* dst_stats;          // the searched stream's stats: [0] largest energy of a pair's span, [1] its centring constant
    int method;                       // SUSHI_HIP_METHOD_CCOEFF_NORMED: spectra of the pattern minus its own mean
    uint4* tspec_low;                 // [segments of the sub-batch][LROWE] the low band again, in bound_low_kernel's order
    float* tnorm_rest;                // [segments of the sub-batch] SQUARED norm of the stored halves outside the band (accumulated: zero it first)
};

template <typename T>
__global__ __launch_bounds__(FT, 8)        // (64 VGPRs: two workgroups of sixteen waves per CU -- at 69 only one fits, 1.15 -> 1.6 ms)
void tspec_kernel(TspecArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    const int tid = threadIdx.x;
    const sushi_fft::Twiddles tw = sushi_fft::load_twiddles<FFT_LOGN, -1>(tid, twiddles());
    const int seg = a.sub_first_seg + blockIdx.x;
    const int k = find_search_by_seg(a.searches, a.n_sub, seg);
    const SearchDesc sd = a.searches[k];
    const int s = seg - sd.first_seg;
    const int M = sd.tmpl_len;
    const TemplStats ts_all = templ_stats(a.src_s1, a.src_s2, sd.tmpl_off, M, a.centre);
    // TM_CCOEFF_NORMED correlates the pattern MINUS ITS OWN MEAN: sum (T - mean T) I is that method's numerator as it is (no
    // window-sum term left to subtract, nothing of the pattern's level in the products), and what the spectra hold -- and the
    // scales are sized by -- is the centred pattern's norm.  (A pattern without variance is answered without its spectra.)
    const bool cc = a.method == SUSHI_HIP_METHOD_CCOEFF_NORMED;
    const double tn_spec = cc && !ts_all.flat ? ts_all.tnorm_c : ts_all.tnorm;
    const float t_sub = cc ? (float)ts_all.tmean : 0.f;
    const float y_scale = y_scale_for(tn_spec, (M + FFT_SEG - 1) / FFT_SEG, a.dst_stats[0]);
    const float t_scale = t_scale_for(tn_spec);
    if (s == 0) {
        const FftLayout lay = fft_layout(sd.win_start, sd.n_pos, M);
        int* __restrict__ pm = a.pairmap + (sd.first_pair - a.sub_first_pair);
        for (int i = tid; i < lay.n_pairs; i += FT) pm[i] = k;
        if (tid == 0) {
            const TemplStats ts = templ_stats(a.src_s1, a.src_s2, sd.tmpl_off, M, a.centre);
            TemplConsts tc;
            tc.tU = ts.tU; tc.inv_tnorm = (float)(1.0 / ts.tnorm); tc.tnorm = (float)ts.tnorm;
            tc.tmean = (float)ts.tmean; tc.flat = ts.flat ? 1 : 0;
            tc.inv_tnorm_c = ts.flat ? 0.f : (float)(1.0 / ts.tnorm_c); tc.inv_m = (float)(1.0 / (double)M);
            tc.c_sum_t = cc ? 0.f : (float)(a.dst_stats[1] * ts.tS1);      // (the centred pattern sums to zero)
            tc.inv_scale = 1.0f / y_scale;
            tc.mac_scale = (float)((double)y_scale / ((double)t_scale * (double)z_scale_for(a.dst_stats[0])));
            a.tconst[k] = tc;
        }
    }
    const T* __restrict__ t = (const T*)a.src_raw + sd.tmpl_off + (int64_t)s * FFT_SEG;
    const int len = min(FFT_SEG, M - s * FFT_SEG);
    cpx v[sushi_fft::PER];
#pragma unroll
    for (int r = 0; r < sushi_fft::PER; ++r) {
        const int e = sushi_fft::in_index<FFT_LOGN>(tid, r);
        const float xa = (float)t[e < len ? e : len - 1] - t_sub;
        v[r].x = e < len ? xa : 0.f;
        v[r].y = 0.f;
    }
    sushi_fft::fft_split<FFT_LOGN, -1>(v, tid, lds, tw);
    const float sc = t_scale / (float)FN;
    // The low entry straight to its place (sixteen half-written lines per wave, all completed by this workgroup within
    // microseconds) and the wave's share of the norm's SQUARE to the segment's accumulator (zeroed before the launch; slb_kernel
    // takes the root): staging both through the LDS, as spectra_kernel does once per stream, cost this kernel -- which runs every
    // step -- two barriers more and 0.6 ms of 1.0 at BASELINE configs[2].  Both leave BEFORE the hand-over: nothing of them stays live.
    {
        float rest2;
        const uint4 low = low_entry_and_rest(v, sc, tid, rest2);
        a.tspec_low[(size_t)blockIdx.x * LROWE + sushi_fft::lslot_of_thread(tid)] = low;
        const float w = wave_sum_shfl(rest2);
        if ((tid & 63) == 0) atomicAdd(a.tnorm_rest + blockIdx.x, w);
    }
    to_load_order(v, tid, lds);
    uint4* __restrict__ out = reinterpret_cast<uint4*>(a.tspec + (size_t)blockIdx.x * FN);
#pragma unroll
    for (int u = 0; u < sushi_fft::PER / 4; ++u)
        out[sushi_fft::wslot_uint4(tid, u)] = uint4{pack_h2(v[4 * u].x * sc, v[4 * u].y * sc), pack_h2(v[4 * u + 1].x * sc, v[4 * u + 1].y * sc),
                                                    pack_h2(v[4 * u + 2].x * sc, v[4 * u + 2].y * sc), pack_h2(v[4 * u + 3].x * sc, v[4 * u + 3].y * sc)};
}

// ------------------------------------------------------------------------------------------
// Frequency-domain multiply-accumulate.  A wave = MAC_SPW searches of one segment-count class, neighbours in the stream (window starts),
// x MAC_BPW entries of four adjacent bins: lane = (search slot, entry).  All searches sit on the same absolute block grid,
// so the lanes of a wave walk the union of their block ranges together (mac_core.hpp): the row piece Z_j(32 bins) is one
// 128-byte line that the eight search slots read at the same address -- one L2 request serves eight searches x 32 bins --
// and every output Y_I(32 bins) of a search is a full 128-byte line written by eight neighbouring lanes.  Pattern spectra
// live in registers (per lane: its own search's, one 32-bit word per bin), a ring of SMAX / FFT_STEP float32 outputs is
// live per lane.  No barriers.
// What bounds the kernel is the bytes a CU's L1 passes (~10 B per clock): with every operand a packed half a byte through
// the L1 feeds twice the multiply-adds it fed as float32, and they are two to an instruction.
// ------------------------------------------------------------------------------------------
constexpr int MAC_SPW = 8;                       // searches per wave
constexpr int MAC_BPW = 64 / MAC_SPW;            // 4-bin entries per wave
constexpr int MAC_WAVES = 4;
constexpr int MAC_THREADS = MAC_WAVES * 64;
constexpr int MAC_BW = MAC_BPW * MAC_WAVES;      // entries per workgroup

struct MacArgs {
    const uint4* spec;                // destination spectra, as 4-bin entries of packed halves
    int64_t spec_blocks;              // blocks of the stream; block `spec_blocks` is all zero
    const uint4* tspec;
    uint4* y;                         // [pairs of the sub-batch][ROWE]
    const SearchDesc* searches;       // the sub-batch's searches
    const TemplConsts* tconst;        // [searches of the sub-batch]: mac_scale
    const int* items;                 // [n_items][1 + MAC_SPW]: segment-count class, then search indices inside the sub-batch (-1 = none)
    int n_items;
    int sub_first_seg;
    int sub_first_pair;
    int chunk_group;                  // bin chunks an XCD works on at a time (a power of two dividing its share)
    uint4* dummy;                     // [MAC_DUMMY_LINES][MAC_THREADS] where the stores of lanes without a valid output go
    const int* enable;                // NULL, or a device flag: 0 = this launch is not needed (every workgroup leaves at once)
};

constexpr int MAC_DUMMY_LINES = 1024;
constexpr int MAC_CHUNKS = ROWE / MAC_BW;
static_assert(MAC_CHUNKS % 8 == 0, "every XCD owns the same number of bin chunks");
constexpr int MAC_ZR = 6;                        // rows per load instruction: divides every SMAX

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(v, d, 64); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return v;
}

constexpr int MAC_SMAX_SHORT = mac_class_smax(MAC_SHORT_CLASSES - 1);   // 18: mac_kernel
constexpr int MAC_SMAX_LONG = mac_class_smax(MAC_CLASSES - 1);           // 30: mac_long_kernel
constexpr int MAC_CH = MAC_ZR;                   // rows one load instruction brings (lane = row x entry)
static_assert(MAC_CH <= MAC_SPW, "a load's rows are spread over the search slots");
constexpr int MAC_AHEAD_ROWS = 36;               // rows in flight per wave (mac_kernel: a multiple of each of its SMAX; mac_long_kernel: one group): 4.5 KB

__device__ __forceinline__ sushi_mac::h8 as_h8(const uint4 v) { return sushi_mac::h8{{v.x, v.y, v.z, v.w}}; }
__device__ __forceinline__ uint4 as_uint4(const sushi_mac::h8 v) { return uint4{v.w[0], v.w[1], v.w[2], v.w[3]}; }

// The walk of one wave (MAC_SPW searches x MAC_BPW entries) for its segment-count class.  Rows reach the lanes in
// two hops: a load instruction fetches MAC_CH consecutive rows at once (lane = (row, entry): 128 distinct bytes per
// row), LA = MAC_AHEAD_ROWS / SMAX groups ahead of their use into a register ring; one group ahead they are dropped
// into the wave's private LDS buffer -- the piece itself and its rotation by -i, which the imaginary parts' dot products
// take (mac_core.hpp): made once here instead of once per search slot -- from where every search slot reads the same row
// (a broadcast read).  The wave's own LDS operations are ordered, so none of this needs a barrier.
// Every memory operation of the loop body is unconditional -- lanes without a row to fetch re-fetch a neighbour's,
// lanes without a valid output store to a dummy line: the compiler then knows how many operations are in flight at every
// point and waits for exactly the load it needs (a conditional one makes it drain everything, every group).
template <int SMAX, bool ACCUM, int ZROWS, int RE>
__device__ __forceinline__ void mac_chunk(const MacArgs& a, const int c0, const int wv_first, const int wv_last,
                                          const long long pair_lo, const long long pair_hi, const bool lane_chunk,
                                          const sushi_mac::h8 (&tt)[SMAX], const uint4* __restrict__ zsp, const int z_zero,
                                          uint4* __restrict__ yout, uint4* __restrict__ dummy, const float sy, const int slot,
                                          const int fb, uint4 (*zw)[ZROWS + 1][2][MAC_BPW]) {
    using sushi_mac::acc4;
    using sushi_mac::zrow;
    constexpr int STEP = FFT_STEP;
    constexpr int AHEAD = SMAX < MAC_SMAX_SHORT ? MAC_AHEAD_ROWS : SMAX;    // rows in flight (the largest class of each kernel: what its registers allow)
    constexpr int NC = SMAX / MAC_CH;                           // load instructions per group
    constexpr int LA = AHEAD / SMAX;                            // groups between a load and its use
    constexpr int NQ = LA * NC;                                 // register ring, in loads
    constexpr int UNROLL = (LA % 2) ? 2 * LA : LA;              // ring slot and LDS buffer of a group are compile-time
    static_assert(SMAX % MAC_CH == 0 && AHEAD % SMAX == 0 && SMAX <= ZROWS, "ring geometry");
    const int lrow = slot % MAC_CH;                              // the row of a load this lane fetches
    const bool loader = slot < MAC_CH;                           // ... and whether its copy is the one that goes to LDS
    // rows jrow + lrow of one load; blocks past the stream are the all-zero block
    auto load_rows = [&](const int jrow) {
        const int jj = jrow + lrow;
        return zsp[(size_t)(jj < z_zero ? jj : z_zero) * RE];
    };
    // a loaded piece into an LDS buffer: the row as it is and rotated
    auto drop = [&](const int buf, const int c, const uint4 piece) {
        const int row = loader ? MAC_CH * c + slot : ZROWS;
        zw[buf][row][0][fb] = piece;
        zw[buf][row][1][fb] = as_uint4(sushi_mac::rot_mi(as_h8(piece)));
    };
    acc4 acc[SMAX / STEP];
#pragma unroll
    for (int r = 0; r < SMAX / STEP; ++r) acc[r] = sushi_mac::zero_acc();
    uint4 rq[NQ];
    // prologue: group 0 straight into LDS buffer 0, groups 1 .. LA into the register ring
    {
        uint4 first[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) first[c] = load_rows(wv_first + c0 + MAC_CH * c);
#pragma unroll
        for (int g = 1; g <= LA; ++g) {
#pragma unroll
            for (int c = 0; c < NC; ++c) rq[(g % LA) * NC + c] = load_rows(wv_first + c0 + SMAX * g + MAC_CH * c);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) drop(0, c, first[c]);
    }
    for (int jb = wv_first; jb <= wv_last; jb += SMAX * UNROLL) {
#pragma unroll
        for (int t = 0; t < UNROLL; ++t) {
            const int jg = jb + SMAX * t;                      // group t of this round (groups past wv_last: dummy work)
            // the next group leaves the register ring for the other LDS buffer (this wave read that buffer one group
            // ago: its LDS operations are in order) ...
#pragma unroll
            for (int c = 0; c < NC; ++c) drop((t + 1) & 1, c, rq[((t + 1) % LA) * NC + c]);
            // ... and its ring slots take the loads of the group LA further on
#pragma unroll
            for (int c = 0; c < NC; ++c) rq[((t + 1) % LA) * NC + c] = load_rows(jg + c0 + SMAX * (1 + LA) + MAC_CH * c);
            auto get_z = [&](const int u) { return zrow{as_h8(zw[t & 1][u][0][fb]), as_h8(zw[t & 1][u][1][fb])}; };
            auto store = [&](const int i, const bool valid, const acc4& v) {
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                // (a lane-predicated store instead of the dummy line was tried: the compiler branches around it and
                // falls back to draining the load queue, tools/experiments/README.md)
                const bool ok = valid && lane_chunk && jg <= wv_last;
                u4* __restrict__ dst = reinterpret_cast<u4*>(ok ? yout + (size_t)i * RE : dummy);
                float re[sushi_mac::BINS], im[sushi_mac::BINS];
#pragma unroll
                for (int k = 0; k < sushi_mac::BINS; ++k) { re[k] = v.re[k] * sy; im[k] = v.im[k] * sy; }   // to the scale of Y
                if (ACCUM) {                                    // patterns beyond one pass: the row accumulates (in halves)
                    // (the entry as EIGHT halves, not `bit_cast<half2>(prev[k])`: hipcc 7.2 reads the first word four times -- add_abs2_entry)
                    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                    static_assert(sushi_mac::BINS == 4, "an entry is four bins");
                    const h8 pv = __builtin_bit_cast(h8, *dst);
#pragma unroll
                    for (int k = 0; k < sushi_mac::BINS; ++k) { re[k] += (float)pv[2 * k]; im[k] += (float)pv[2 * k + 1]; }
                }
                u4 o;
#pragma unroll
                for (int k = 0; k < sushi_mac::BINS; ++k) {
                    const h2 q = {(_Float16)re[k], (_Float16)im[k]};                 // v_cvt_pk_f16_f32: round to nearest even
                    o[k] = __builtin_bit_cast(unsigned, q);
                }
                // Y is streamed once and read back once by another kernel: non-temporal stores keep the block spectra in
                // L2.  A store with dummy lanes in it goes the write-back way instead (one store instruction on either
                // path): the dummy lines are overwritten in L2 again and again and never reach HBM, whereas non-temporal
                // stores to them would all be written through (measured: 68 GB of writes per launch for 46 GB of Y).
                if (__ballot(ok) == ~0ull) __builtin_nontemporal_store(o, dst);
                else *dst = o;
            };
            // (a row of the large classes meets three or more segments: one row of look-ahead covers the LDS latency, and
            // their registers do not hold two)
            sushi_mac::mac_group<SMAX, STEP, (SMAX >= MAC_SMAX_SHORT ? 2 : 3)>((long long)jg, pair_lo, pair_hi, tt, acc, get_z, store);
        }
    }
}

template <int SMAX, int ZROWS, int RE>
__device__ __forceinline__ void mac_item(const MacArgs& a, const int* __restrict__ item, const int e0, const int slot,
                                         uint4 (*zw)[ZROWS + 1][2][MAC_BPW]) {
    using sushi_mac::h8;
    constexpr int STEP = FFT_STEP;
    const int fb = threadIdx.x % MAC_BPW;
    const int k = item[1 + slot];                               // this lane's search
    long long pair_lo = 0, pair_hi = 0;
    int jb0 = 0x7fffffff, jb1 = -0x7fffffff, n_seg = 0, first_seg = 0, first_pair = 0;
    float sy = 0.f;
    if (k >= 0) {
        const SearchDesc sd = a.searches[k];
        const FftLayout lay = fft_layout(sd.win_start, sd.n_pos, sd.tmpl_len);
        pair_lo = lay.pair0; pair_hi = lay.pair0 + lay.n_pairs; n_seg = lay.n_seg;
        first_seg = sd.first_seg - a.sub_first_seg; first_pair = sd.first_pair - a.sub_first_pair;
        sy = a.tconst[k].mac_scale;
        long long g0, g1;
        sushi_mac::group_range<SMAX, STEP>(pair_lo, pair_hi, &g0, &g1);
        jb0 = (int)g0; jb1 = (int)g1;                            // block indices fit 31 bits (sushi_hip_stream_add_spectra)
    }
    // the wave walks the union of its lanes' group ranges (wave-uniform loop bounds)
    const int wv_first = __builtin_amdgcn_readfirstlane(wave_min_i32(jb0));
    const int wv_last = __builtin_amdgcn_readfirstlane(wave_max_i32(jb1));
    const int wv_seg = __builtin_amdgcn_readfirstlane(wave_max_i32(n_seg));
    const uint4* __restrict__ tsp = a.tspec + (size_t)first_seg * RE + e0;
    uint4* __restrict__ yout = a.y + (size_t)first_pair * RE + e0;
    const uint4* __restrict__ zsp = a.spec + e0;
    // one 16-byte slot per wave: the invalid lanes of a store instruction then add one request to it instead of a line per
    // search slot (a fifth of mac_kernel's write requests were dummy lines, and the CU's L1 write path is what it waits for)
    uint4* __restrict__ dummy = a.dummy + (size_t)(blockIdx.x % MAC_DUMMY_LINES) * MAC_THREADS + (threadIdx.x & ~63);
    const int z_zero = (int)(a.spec_blocks < 0x7fffffff ? a.spec_blocks : 0x7fffffff);
    for (int c0 = 0; c0 < wv_seg; c0 += SMAX) {                 // patterns longer than SMAX segments: SMAX at a time
        h8 tt[SMAX];
#pragma unroll
        for (int s = 0; s < SMAX; ++s) tt[s] = (c0 + s) < n_seg ? as_h8(tsp[(size_t)(c0 + s) * RE]) : sushi_mac::zero_h8();
        const bool lane_chunk = c0 < n_seg;
        if (c0 == 0) mac_chunk<SMAX, false, ZROWS, RE>(a, c0, wv_first, wv_last, pair_lo, pair_hi, lane_chunk, tt, zsp, z_zero, yout, dummy, sy, slot, fb, zw);
        else if (SMAX == MAC_SMAX_LONG) mac_chunk<SMAX, true, ZROWS, RE>(a, c0, wv_first, wv_last, pair_lo, pair_hi, lane_chunk, tt, zsp, z_zero, yout, dummy, sy, slot, fb, zw);
    }
}

// Grid: MAC_CHUNKS bin chunks x items (in the order of their windows in the stream).  Workgroup b runs on XCD b % 8 (observed; speed only): every XCD
// owns MAC_CHUNKS / 8 bin chunks, takes them `chunk_group` at a time and walks the items in stream order for each
// group, so that the workgroups in flight on an XCD are the same few chunks of neighbouring items, whose windows
// overlap: a row fetched for one is found in that XCD's L2 by the others.
template <int RE>
__device__ __forceinline__ void mac_place(const MacArgs& a, int* item_idx, int* e0, int* slot, int* wave) {
    constexpr int CPX = RE / MAC_BW / 8;                        // chunks per XCD
    const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;
    const int cg = a.chunk_group;
    const int per_group = cg * a.n_items;
    const int grp = kx / per_group;
    const int in_grp = kx - grp * per_group;
    *item_idx = in_grp / cg;
    const int chunk = xcd * CPX + grp * cg + (in_grp - *item_idx * cg);
    const int lane = threadIdx.x & 63;
    *wave = threadIdx.x >> 6;
    *e0 = chunk * MAC_BW + *wave * MAC_BPW + (lane % MAC_BPW);   // which entry of four bins
    *slot = lane / MAC_BPW;                                      // which of the item's searches
}

// RE = entries of a row: ROWE (whole spectra) or LROWE (the low-band rows of the band-split exclusion: the same walk over a
// quarter of the bins -- rows, pattern spectra and products only have to agree on one order of the entries).
template <int RE>
__global__ __launch_bounds__(MAC_THREADS, 3)
void mac_kernel(MacArgs a) {
    __shared__ uint4 zring[MAC_WAVES][2][MAC_SMAX_SHORT + 1][2][MAC_BPW];   // per wave: two groups of rows, each with its rotation (+ a row nobody reads)
    if (a.enable && *a.enable == 0) return;
    int item_idx, e0, slot, wave;
    mac_place<RE>(a, &item_idx, &e0, &slot, &wave);
    const int* __restrict__ item = a.items + (size_t)item_idx * (1 + MAC_SPW);
    switch (item[0]) {                                          // class c holds patterns of up to 6 (c + 1) segments
        case 0: mac_item<6, MAC_SMAX_SHORT, RE>(a, item, e0, slot, zring[wave]); break;
        case 1: mac_item<12, MAC_SMAX_SHORT, RE>(a, item, e0, slot, zring[wave]); break;
        default: mac_item<18, MAC_SMAX_SHORT, RE>(a, item, e0, slot, zring[wave]); break;
    }
}

// Patterns of 19 .. 30 segments (and, 30 at a time, longer ones): up to 30 pattern spectra per lane, two waves per SIMD.
// One pass instead of mac_kernel's two with Y read back in between (BASELINE configs[4]: half of the events).
template <int RE>
__global__ __launch_bounds__(MAC_THREADS, 2)
void mac_long_kernel(MacArgs a) {
    __shared__ uint4 zring[MAC_WAVES][2][MAC_SMAX_LONG + 1][2][MAC_BPW];
    if (a.enable && *a.enable == 0) return;
    int item_idx, e0, slot, wave;
    mac_place<RE>(a, &item_idx, &e0, &slot, &wave);
    const int* __restrict__ item = a.items + (size_t)item_idx * (1 + MAC_SPW);
    switch (item[0]) {
        case 3: mac_item<24, MAC_SMAX_LONG, RE>(a, item, e0, slot, zring[wave]); break;
        default: mac_item<30, MAC_SMAX_LONG, RE>(a, item, e0, slot, zring[wave]); break;
    }
}

// The multiply-accumulate of LISTED pairs (the band-split exclusion: the pairs transformed first and the pairs the bound
// could not exclude -- a few per cent of all), over whole rows.  A workgroup = 256 consecutive entries of one listed pair,
// a lane one entry: its sum over the pattern's segments in mac_kernel's own order and arithmetic (patterns beyond
// MAC_SMAX_LONG segments: that many per pass, the row re-rounded to halves in between, as mac_long_kernel leaves it).
struct MacListArgs {
    const uint4* spec;
    int64_t spec_blocks;
    const uint4* tspec;
    uint4* y;                         // [pairs of the sub-batch][ROWE]
    const SearchDesc* searches;
    const TemplConsts* tconst;
    const int* pairmap;
    const int* list;                  // pairs to compute (indices inside the sub-batch)
    const int* count;                 // NULL, or how many entries of `list` exist
    int n_list;                       // entries of `list` when count is NULL
    int sub_first_seg;
    int sub_first_pair;
    const int* disable;               // NULL, or a device flag: 1 = the dense multiply-accumulate forms every row instead
    int long_only;                    // 1: only pairs of patterns beyond MAC_SMAX_LONG segments (mac_rows_kernel forms the others)
};
constexpr int MACL_THREADS = 256;
constexpr int MACL_PARTS = ROWE / MACL_THREADS;
__global__ __launch_bounds__(MACL_THREADS)
void mac_list_kernel(MacListArgs a) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    if (a.disable && *a.disable) return;
    const int n = a.count ? *a.count : a.n_list;
    const int z_zero = (int)(a.spec_blocks < 0x7fffffff ? a.spec_blocks : 0x7fffffff);
    for (long long it = blockIdx.x; it < (long long)n * MACL_PARTS; it += gridDim.x) {
        const int pr = a.list[it / MACL_PARTS];
        const int e = (int)(it % MACL_PARTS) * MACL_THREADS + threadIdx.x;
        const int k = a.pairmap[pr];
        const SearchDesc sd = a.searches[k];
        const FftLayout lay = fft_layout(sd.win_start, sd.n_pos, sd.tmpl_len);
        if (a.long_only && lay.n_seg <= MAC_SMAX_LONG) continue;
        const long long I = lay.pair0 + (a.sub_first_pair + pr - sd.first_pair);
        const float sy = a.tconst[k].mac_scale;
        const uint4* __restrict__ tsp = a.tspec + (size_t)(sd.first_seg - a.sub_first_seg) * ROWE + e;
        const uint4* __restrict__ zsp = a.spec + e;
        float re[sushi_mac::BINS], im[sushi_mac::BINS];
        for (int c0 = 0; c0 < lay.n_seg; c0 += MAC_SMAX_LONG) {
            sushi_mac::acc4 acc = sushi_mac::zero_acc();
            const int s_end = c0 + MAC_SMAX_LONG < lay.n_seg ? c0 + MAC_SMAX_LONG : lay.n_seg;
            // six segments at a time, all twelve loads requested before the first product (a segment past the end multiplies a
            // zero pattern entry: the sum is unchanged, as in the padded classes of mac_kernel)
            for (int s6 = c0; s6 < s_end; s6 += 6) {
                sushi_mac::h8 z[6], u[6];
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    const int sx = s6 + t < s_end ? s6 + t : s_end - 1;
                    const long long jj = FFT_STEP * I + sx;
                    z[t] = as_h8(zsp[(size_t)(jj < z_zero ? jj : z_zero) * ROWE]);
                    u[t] = as_h8(tsp[(size_t)sx * ROWE]);
                }
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    const sushi_mac::h8 ut = s6 + t < s_end ? u[t] : sushi_mac::zero_h8();
                    const sushi_mac::zrow zr = {z[t], sushi_mac::rot_mi(z[t])};
                    sushi_mac::mac4(acc, ut, zr);
                }
            }
            unsigned o[sushi_mac::BINS];
#pragma unroll
            for (int q = 0; q < sushi_mac::BINS; ++q) {
                float r = acc.re[q] * sy, i = acc.im[q] * sy;
                if (c0 > 0) { r += re[q]; i += im[q]; }
                const h2 h = {(_Float16)r, (_Float16)i};
                o[q] = __builtin_bit_cast(unsigned, h);
                re[q] = (float)h.x; im[q] = (float)h.y;                      // what a later pass reads back
            }
            if (s_end == lay.n_seg) a.y[(size_t)pr * ROWE + e] = uint4{o[0], o[1], o[2], o[3]};
        }
    }
}

// ------------------------------------------------------------------------------------------
// Inverse transform of one block pair + fused scoring epilogue
// ------------------------------------------------------------------------------------------
struct IfftArgs {
    const uint2* y;                   // [pairs of the sub-batch][FN / 2]: the products as packed halves
    const double* dst_stats;          // [1]: the constant the block spectra are centred by
    const SearchDesc* searches;       // the sub-batch's searches
    int n_sub;
    int first_search;                 // global index of searches[0]
    int sub_first_pair;
    int64_t dst_len;
    float delta;
    unsigned long long* cand;         // [pairs of the sub-batch][FFT_ROW]
    float* pair_lb;                   // [pairs of the sub-batch] smallest lower bound (score - e) of the pair: what refine_kernel scans
    unsigned long long* gkeys;        // [all searches] running minimum of (f32 score + error bound)
    const int* pairmap;               // [pairs of the sub-batch] -> search index inside the sub-batch
    const int* order;                 // [pairs of the sub-batch] workgroup -> pair (L2-friendly schedule; or a list of pairs), or NULL
    const int* count;                 // NULL, or how many entries of `order` exist: workgroups beyond leave at once
    const TemplConsts* tconst;        // [searches of the sub-batch]
    const float* urel;                // dst stream: prefix of the uncentred squares relative to its block's base
    const double* ubase;              // dst stream: those block bases [nb + 1]
    const float* usrel;               // dst stream: (urel, prefix of the samples relative to its block's base) interleaved (TM_CCOEFF_NORMED)
    const double* sbase;              // dst stream: those block bases [nb + 1]
    int64_t nb;                       // blocks of the dst stream
    // collection pass only
    const int* flags;                 // [all searches] 1 = list the candidates, 2 = every position
    const int* flag_list;             // flagged searches of this sub-batch (global indices)
    const int* sub_flagged;           // how many
    const int* citems;                // the (flagged search, pair) items refine_kernel listed: pair indices inside the sub-batch
    const int* n_citems;              // [1] how many
    TileDesc* tiles;
    int32_t* candbuf;
    int cand_cap;
    RunCounters* counters;
    // the audit of the exclusion: every transformed pair's lower bound against what the pair really scores
    const float* slb;                 // [pairs of the sub-batch] or NULL (no exclusion in this run)
    const unsigned char* audit_mark;  // [pairs of the sub-batch] bit 0 = the bound had EXCLUDED this pair (transformed as a check)
    int* viol;                        // [all searches] set to 1 where a lower bound turns out above a real score
    int list_first;                   // with `count`: the first list slot this launch takes ...
    int list_direct;                  // ... one workgroup per slot (ifft_kernel), or a fixed grid striding from there on (ifft_list_kernel)
};

constexpr int GQ = 4;                      // positions per window-energy load group
static_assert(GQ % 2 == 0 && HPT % (GQ / 2) == 0, "a load group is GQ / 2 positions of both halves");

// Everything the epilogue of one pair needs, per thread.  Scores are indexed half * HPT + r for position
// pos = half * FH + tid + FT * r of the pair.  A score is what the ranking MINIMISES: the TM_SQDIFF_NORMED value itself,
// or 1 - the TM_CCOEFF_NORMED value (arg-max as an arg-min, in [0, 2]).  Positions outside the search hold +inf;
// TM_CCOEFF_NORMED positions whose window variance is inside its own f32 rounding error ("uncertain": flat or nearly
// flat windows, whose score the f32 stage cannot bound) hold UNCERTAIN: always candidates, never part of the pair's minimum.
constexpr float UNCERTAIN = -1.0f;
struct PairScores {
    float scores[2 * HPT];
    float best;          // smallest score of this thread (uncertain positions aside)
    float max_rs;        // largest 1 / sqrt(window energy | window variance sum) over this thread's valid, certain positions
    int any_uncertain;
};

// constants of the TM_CCOEFF_NORMED error model, in units of eps = 2^-24:
//   numerator   sum (T - mean T) I : |err| <= eps * |T_c| * |Z| * FFT_KE   (the cross term of the centred pattern as it is)
//   variance    wU - wS1^2 / M     : |err| <= eps * |Z|^2 * CD,  CD = 28 + 512 / sqrt(M)
// (|Z|^2 = the energy of the samples that enter the pair's transforms; window sums come from float32 prefix values
// relative to per-block float64 bases: each of them is off by <= eps * 64 |Z|, wS1 <= sqrt(M) |Z|, tmean <= |T| / sqrt(M)).
__device__ __forceinline__ float ccoeff_cd(float inv_sqrt_m) { return 28.0f + 512.0f * inv_sqrt_m; }

// Y of one pair (packed halves) into the registers of the inverse transform.  Y is stored in the order the transform loads
// it: one 16-byte load brings four registers, a wave's load instruction one contiguous KiB.  Issued before anything else a
// workgroup does: the address needs the pair index only, and the search's descriptor and constants (two more dependent
// loads) are not needed before the epilogue -- with two workgroups per CU every serial hop at a workgroup's start is CU time.
__device__ __forceinline__ float load_y(sushi_fft::uint4v (&yl)[4], const uint2* __restrict__ yin, const int tid) {
    const sushi_fft::uint4v* __restrict__ yh = reinterpret_cast<const sushi_fft::uint4v*>(yin);
    float q2 = 0.f;                                              // energy of this thread's part of the row (quantisation model)
#pragma unroll
    for (int u = 0; u < sushi_fft::PER / 4; ++u) {
        yl[u] = yh[sushi_fft::wslot_uint4(tid, u)];              // four bins: eight halves, as they go to the matrix pipe
        q2 = add_abs2_entry(yl[u], q2);                          // |Y(f)|^2 straight from the halves
    }
    return q2;
}

// One pair: inverse transform of the loaded Y, f32 scores.  Returns through `ps`; `plo`/`phi` bound the valid positions.
// METHOD 0  score = sum (T - I)^2 / sqrt(sum T^2 * sum I^2), everything UNCENTRED: the f32 error of the FFT'd cross term
//           is then bounded by FFT_KE * eps * |T| * |Z| (sushi_common.hpp), i.e. in score units 2 * FFT_KE * eps * |Z| / |window|.
// METHOD 1  score = 1 - (sum T I - sum I * mean T) / sqrt(sum (T - mean T)^2 * (sum I^2 - (sum I)^2 / M)): the same cross
//           term with the same bound; the window sums come from the stream's second relative prefix (srel / sbase).
template <int METHOD>
__device__ __forceinline__ void score_pair(const IfftArgs& a, const sushi_fft::uint4v (&yl)[4], const sushi_fft::MfmaB& mb, const SearchDesc& sd,
                                           const TemplConsts& tc, const int64_t pairI, float* lds, const int tid,
                                           const sushi_fft::WTwiddles& tw, PairScores& ps, int& plo_out, int& phi_out,
                                           float& zn_out, float& znc_out) {
    constexpr bool CC = METHOD == SUSHI_HIP_METHOD_CCOEFF_NORMED;
    constexpr int GQM = CC ? GQ / 2 : GQ;                        // METHOD 1 holds two values per window end: half as many positions per group
    constexpr int NG = 2 * HPT / GQM;
    const int M = sd.tmpl_len;
    const int64_t P = sd.n_pos;
    const int64_t w = sd.win_start;
    const int64_t n = a.dst_len;
    const int64_t kA = pairI * FFT_STEP;                         // first block of the pair
    const int64_t qbase = kA * FFT_SEG;                          // dst sample of this pair's first position
    // Window energies come from the stream's block-relative prefix of squares: two 4-byte loads per
    // position (urel[q], urel[q + M]), requested group by group just ahead of the scoring of that group.
    // Positions qbase + pos: scalar base + 32-bit lane offsets; offsets past the end of the stream are
    // clamped so that the loads stay unconditional (such positions are masked below).
    const float* __restrict__ r0b = a.urel + qbase;
    const int64_t room = n - qbase;                                      // urel has n + 1 entries
    const int lim0 = (int)(room < 2 * FH - 1 ? room : 2 * FH - 1);
    const int64_t roomM = room - M;
    const int limM = (int)(roomM < 2 * FH - 1 ? roomM : 2 * FH - 1);     // may be negative
    const float* __restrict__ rMb = limM < 0 ? r0b : r0b + M;
    const int limMc = limM < 0 ? 0 : limM;
    typedef float f2v __attribute__((ext_vector_type(2)));
    const f2v* __restrict__ s0b = reinterpret_cast<const f2v*>(a.usrel) + qbase;      // (METHOD 1) both prefixes, interleaved: one 8-byte load per end
    const f2v* __restrict__ sMb = limM < 0 ? s0b : s0b + M;
    float ra[GQM], rb[GQM], ra_n[GQM], rb_n[GQM];
    float sa[CC ? GQM : 1], sb_[CC ? GQM : 1], sa_n[CC ? GQM : 1], sb_n[CC ? GQM : 1];
    unsigned after_fft = 0;
    const int Mh = M / FFT_SEG, Ml = M - Mh * FFT_SEG;
    // A pair whose 2H positions are all result positions of the search and all loadable (every pair but
    // the first and last of a window, and those at the end of the stream) takes the `interior` variants:
    // immediate load offsets, no clamps, no validity masks.
    const int64_t plo64 = w - qbase;
    const int plo = (int)(plo64 < -1 ? -1 : plo64);              // valid positions: plo <= pos < phi
    const int64_t phi64 = P + (w - qbase);
    const int phi = (int)(phi64 < 2 * FH ? phi64 : 2 * FH);
    const bool interior = lim0 >= 2 * FH - 1 && limM >= 2 * FH - 1 && plo <= 0 && phi >= 2 * FH;
    plo_out = plo; phi_out = phi;
    // Load group g = positions r = (GQM/2) g .. (GQM/2) g + GQM/2 - 1 of BOTH halves: once a group is scored, the transform
    // outputs it used are dead (v[r].x is half 0's cross term, v[r].y half 1's).
    auto load_group = [&](const int g, float (&xa)[GQM], float (&xb)[GQM], float (&ya)[CC ? GQM : 1], float (&yb)[CC ? GQM : 1],
                          auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
        for (int q = 0; q < GQM; ++q) {
            const int r = g * (GQM / 2) + q / 2;
            const int half = q & 1;
            const int pos = tid + FT * r + half * FH;
            if (INTERIOR) {
                // each load's base is an opaque scalar: left visible, base + lane offset is shared between the loads and
                // every displacement past the instruction's 4 KB immediate costs a 64-bit vector add
                if (CC) {
                    typedef const __attribute__((address_space(1))) f2v* gptr2;
                    gptr2 qa = (gptr2)(s0b + FT * r + half * FH);
                    gptr2 qb = (gptr2)(sMb + FT * r + half * FH);
                    asm volatile("" : "+s"(qa));
                    asm volatile("" : "+s"(qb));
                    const f2v va = qa[(unsigned)(tid + after_fft) & 0x7fffu];
                    const f2v vb = qb[(unsigned)(tid + after_fft) & 0x7fffu];
                    xa[q] = va.x; ya[q] = va.y;
                    xb[q] = vb.x; yb[q] = vb.y;
                } else {
                    typedef const __attribute__((address_space(1))) float* gptr;
                    gptr pa = (gptr)(r0b + FT * r + half * FH);
                    gptr pb = (gptr)(rMb + FT * r + half * FH);
                    asm volatile("" : "+s"(pa));
                    asm volatile("" : "+s"(pb));
                    xa[q] = pa[(unsigned)(tid + after_fft) & 0x7fffu];
                    xb[q] = pb[(unsigned)(tid + after_fft) & 0x7fffu];
                }
            } else {
                const unsigned oa = (((unsigned)(pos <= lim0 ? pos : lim0)) + after_fft) & 0x7fffu;
                const unsigned ob = (((unsigned)(pos <= limM ? pos : limMc)) + after_fft) & 0x7fffu;
                if (CC) {
                    const f2v va = s0b[oa], vb = sMb[ob];
                    xa[q] = va.x; ya[q] = va.y;
                    xb[q] = vb.x; yb[q] = vb.y;
                } else {
                    xa[q] = r0b[oa];
                    xb[q] = rMb[ob];
                }
            }
        }
    };
    // The block bases sit at wave-uniform addresses: scalar loads into SGPRs, issued here and needed after the
    // transform.  Starts: blocks kA .. kA + 2 VB - 1; ends: kA + Mh .. kA + Mh + 2 VB; and the end of the span of
    // samples that enters this pair's transforms (error bound).
    double sb[2 * FFT_VB], eb[2 * FFT_VB + 1];
    double sb1[CC ? 2 * FFT_VB : 1], eb1[CC ? 2 * FFT_VB + 1 : 1];
    double span_end, span_start1, span_end1;                     // (sums of the samples over the span: its centred energy)
    const int n_seg = (M + FFT_SEG - 1) / FFT_SEG;
    if (kA + n_seg + 2 * FFT_VB <= a.nb) {                       // away from the end of the stream: consecutive entries, wide loads
#pragma unroll
        for (int c = 0; c < 2 * FFT_VB; ++c) sb[c] = a.ubase[kA + c];
#pragma unroll
        for (int c = 0; c < 2 * FFT_VB + 1; ++c) eb[c] = a.ubase[kA + Mh + c];
        span_end = a.ubase[kA + n_seg + 2 * FFT_VB];
        span_start1 = a.sbase[kA];
        span_end1 = a.sbase[kA + n_seg + 2 * FFT_VB];
        if (CC) {
#pragma unroll
            for (int c = 0; c < 2 * FFT_VB; ++c) sb1[c] = a.sbase[kA + c];
#pragma unroll
            for (int c = 0; c < 2 * FFT_VB + 1; ++c) eb1[c] = a.sbase[kA + Mh + c];
        }
    } else {
#pragma unroll
        for (int c = 0; c < 2 * FFT_VB; ++c) {
            int64_t bi = kA + c;
            bi = bi < a.nb ? bi : a.nb;
            sb[c] = a.ubase[bi];
            if (CC) sb1[c] = a.sbase[bi];
        }
#pragma unroll
        for (int c = 0; c < 2 * FFT_VB + 1; ++c) {
            int64_t bi = kA + Mh + c;
            bi = bi < a.nb ? bi : a.nb;
            eb[c] = a.ubase[bi];
            if (CC) eb1[c] = a.sbase[bi];
        }
        int64_t bi = kA + n_seg + 2 * FFT_VB;
        bi = bi < a.nb ? bi : a.nb;
        span_end = a.ubase[bi];
        span_end1 = a.sbase[bi];
        int64_t b0 = kA < a.nb ? kA : a.nb;
        span_start1 = a.sbase[b0];
    }
    cpx v[sushi_fft::PER];
    sushi_fft::fft_wave_mfma<1>(yl, v, tid, lds, tw, mb);
    // keep the window loads below the last pass: hoisted above it (the scheduler's preference) they do not
    // fit the register budget next to the radix-16 butterflies and get spilled to scratch one by one.
    // `after_fft` is an opaque zero that the asm "computes" from the needed outputs; added to the load
    // offsets it makes the loads depend on the finished transform.
    // (an asm statement takes at most 30 operands: the outputs are chained through in groups of 4)
#pragma unroll
    for (int g = 0; g < HPT; g += 4) {
        asm volatile("v_mov_b32 %0, 0"
                     : "=v"(after_fft)
                     : "v"(v[g + 0].x), "v"(v[g + 0].y), "v"(v[g + 1].x), "v"(v[g + 1].y), "v"(v[g + 2].x), "v"(v[g + 2].y),
                       "v"(v[g + 3].x), "v"(v[g + 3].y), "v"(after_fft));
    }
    // wave-uniform floats made from float64 arithmetic live in VGPRs unless told otherwise: pin them to SGPRs
    auto uniform = [](const float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
    // window energy = ub0[h][b] (+ dub[h][b] if the window's end falls one block further) + (urel[q + M] - urel[q])
    float ub0[2][FFT_VB], dub[2][FFT_VB];
    float us0[CC ? 2 : 1][FFT_VB], dus[CC ? 2 : 1][FFT_VB];      // (METHOD 1) the same for the window sums
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int b = 0; b < FFT_VB; ++b) {
            const float e0 = (float)(eb[h * FFT_VB + b] - sb[h * FFT_VB + b]);       // window end's base - start's
            const float e1 = (float)(eb[h * FFT_VB + b + 1] - sb[h * FFT_VB + b]);
            ub0[h][b] = uniform(e0);
            dub[h][b] = uniform(e1 - e0);
            if (CC) {
                const float f0 = (float)(eb1[h * FFT_VB + b] - sb1[h * FFT_VB + b]);
                const float f1 = (float)(eb1[h * FFT_VB + b + 1] - sb1[h * FFT_VB + b]);
                us0[h][b] = uniform(f0);
                dus[h][b] = uniform(f1 - f0);
            }
        }
    }
    const float zn = uniform((float)sqrt(fmax(span_end - sb[0], 0.0)));
    zn_out = zn;
    {
        // energy of the CENTRED samples that enter this pair's transforms (what the f32 error of the cross term scales with)
        const int64_t s_lo = qbase < n ? qbase : n;
        const int64_t s_hi64 = (kA + n_seg + 2 * FFT_VB) * (int64_t)FFT_SEG;
        const int64_t s_hi = s_hi64 < n ? s_hi64 : n;
        const double c = a.dst_stats[1];
        const double e2 = (span_end - sb[0]) - 2.0 * c * (span_end1 - span_start1) + c * c * (double)(s_hi - s_lo);
        znc_out = uniform((float)sqrt(fmax(e2, 0.0)) * 1.0000005f);
    }
    // sum T I = y' / scale + c sum T  (block spectra of the centred stream, pattern spectra scaled: module header)
    const float tU = uniform((float)(tc.tU - 2.0 * (double)tc.c_sum_t));
    const float m2s = uniform(-2.0f * tc.inv_scale);
    const float inv_s = uniform(tc.inv_scale);
    const float inv_tnorm = uniform(CC ? tc.inv_tnorm_c : tc.inv_tnorm);
    const float neg_inv_m = uniform(-tc.inv_m);
    // (METHOD 1) a window variance sum below this is inside its own rounding error: four times the modelled error
    const float tau = uniform(4.0f * 5.9604645e-8f * zn * zn * ccoeff_cd(sqrtf(tc.inv_m)));

    // ---- f32 scores (ranking only: the near-minimum ones are re-evaluated exactly) ----
    // METHOD 0: score = num / (sqrt(wU) * tnorm), and 1 where wU <= 0 (cv2's `t = 0` case: the test
    // diff2 <= min(0.5, 10*eps*wU) can only hold for wU <= 0).  cv2's clamp score <= 1 is applied here too:
    // it creates exact ties at 1.0 (no-match windows), and ties must all become candidates for the lowest
    // index to win.
    const int carry_from = FFT_SEG - Ml - tid;                   // window end one block further iff FT*(r % RPB) >= this
    float best_s = __builtin_inff();
    float max_rs = 0.f;
    int any_unc = 0;
    auto score_all = [&](auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        load_group(0, ra, rb, sa, sb_, interior_tag);
#pragma unroll
        for (int g = 0; g < NG; ++g) {                      // (ties are settled by position afterwards)
            if (g + 1 < NG) load_group(g + 1, ra_n, rb_n, sa_n, sb_n, interior_tag);
#pragma unroll
            for (int q = 0; q < GQM; ++q) {
                const int r = g * (GQM / 2) + q / 2;
                const int half = q & 1;
                const int blk = r / RPB;
                const int pos = tid + FT * r + half * FH;
                const bool carry = FT * (r % RPB) >= carry_from;
                const float wU = (ub0[half][blk] + (carry ? dub[half][blk] : 0.f)) + (rb[q] - ra[q]);   // sum I^2
                const float yv = half ? v[r].y : v[r].x;
                float score, rs;
                bool certain;
                if (CC) {
                    const float wS = (us0[half][blk] + (carry ? dus[half][blk] : 0.f)) + (sb_[q] - sa[q]);   // sum I
                    const float num = yv * inv_s;                                    // sum (T - mean T) I: the spectra are of the centred pattern
                    const float d2 = __builtin_fmaf(wS * neg_inv_m, wS, wU);         // sum I^2 - (sum I)^2 / M
                    certain = d2 > tau;
                    rs = __builtin_amdgcn_rsqf(d2);
                    const float cc = __builtin_amdgcn_fmed3f(num * rs * inv_tnorm, -1.0f, 1.0f);
                    score = certain ? 1.0f - cc : UNCERTAIN;
                } else {
                    const float num = __builtin_fmaf(yv, m2s, tU + wU);              // sum (T - I)^2 = sum T^2 + sum I^2 - 2 sum T I
                    // a window without energy scores 1 (cv2's t = 0 case): clamped to a tiny energy its score is huge before
                    // the clamp to 1, and its 1/|window| blows the pair's bound up so that every position of the pair goes to
                    // the exact stages -- right, and one v_max instead of a compare and two selects per position (a pattern
                    // without energy, where num can be 0 as well, is handled before the loop)
                    rs = __builtin_amdgcn_rsqf(fmaxf(wU, 1e-30f));
                    score = num * rs * inv_tnorm;                                    // ~1 ulp: this stage only ranks
                    score = __builtin_amdgcn_fmed3f(score, 0.0f, 1.0f);              // both clamps (keys need score >= 0)
                    certain = true;
                }
                bool valid = true;
                if (!INTERIOR) {
                    valid = pos >= plo && pos < phi;
                    score = valid ? score : __builtin_inff();
                }
                ps.scores[half * HPT + r] = score;
                if (CC) {
                    best_s = fminf(best_s, score >= 0.f ? score : __builtin_inff());
                    any_unc |= (valid && !certain) ? 1 : 0;
                } else {
                    best_s = fminf(best_s, score);
                }
                max_rs = fmaxf(max_rs, (certain && valid) ? rs : 0.f);
            }
            if (g + 1 < NG) {
#pragma unroll
                for (int q = 0; q < GQM; ++q) {
                    ra[q] = ra_n[q]; rb[q] = rb_n[q];
                    if (CC) { sa[q] = sa_n[q]; sb_[q] = sb_n[q]; }
                }
            }
        }
    };
    if (interior) score_all(std::true_type()); else score_all(std::false_type());
    if (!CC && !(tc.tU > 0.0)) {
        // a pattern of zeros: cv2's result is all ones (t = 0 everywhere), while num * rs * (1 / 0) above is NaN where the window
        // has no energy either
#pragma unroll
        for (int q = 0; q < 2 * HPT; ++q) ps.scores[q] = ps.scores[q] < __builtin_inff() || ps.scores[q] != ps.scores[q] ? 1.0f : ps.scores[q];
        best_s = 1.0f;
        max_rs = fmaxf(max_rs, 1e15f);
    }
    ps.best = best_s;
    ps.max_rs = max_rs;
    ps.any_uncertain = any_unc;
}

// error bound of a pair in score units (module header, sushi_common.hpp FFT_KE).
// METHOD 0: the cross term's 2 * KE * eps * |Z| / |window| plus the window energy's own f32 rounding,
//           16 * eps * |Z|^2 / (|T| * |window|)
// METHOD 1: numerator error / (|T_c| |W_c|) + the relative error of 1 / sqrt(variance sum) (the score is <= 1 in size)
template <int METHOD>
__device__ __forceinline__ float pair_error_model(float zn, float zn_c, float max_rs, const TemplConsts& tc, float q2,
                                                  float inv_scale, int mac_passes) {
    const float eps = 5.9604645e-8f;                             // 2^-24
    // quantisation of the Y row: every stored half is off by <= 2^-11 of its size (round to nearest), independently; the
    // inverse transform sums N of them: variance (2^-22 / 3) * sum |Y(f)|^2 (+ the subnormal floor), Y_KQ deviations
    // (2^-22 / 3) x (the row's own rounding, once per accumulating pass of mac_long_kernel: a pattern of more than MAC_SMAX_LONG
    // segments re-rounds the partial row every pass; + its two factors' roundings: module header)
    const float sigma_y = sqrtf(q2 * (7.9472862e-8f * (float)(2 + mac_passes)) + (float)FN * 1.2e-15f) * inv_scale;
    if (METHOD == SUSHI_HIP_METHOD_CCOEFF_NORMED) {
        const float ism = sqrtf(tc.inv_m);
        const float z = fmaxf(zn, zn_c);
        // (numerator: the cross term of the CENTRED pattern, FFT_KE eps |T_c| |Z|, over |T_c| |W_c|)
        return eps * max_rs * z * (FFT_KE + ccoeff_cd(ism) * z * max_rs) + Y_KQ * sigma_y * max_rs * tc.inv_tnorm_c;
    }
    return eps * max_rs * (2.0f * FFT_KE * zn_c + 16.0f * zn * zn * tc.inv_tnorm) + 2.0f * Y_KQ * sigma_y * max_rs * tc.inv_tnorm;
}

// sum over the workgroup of a per-thread value, the same in ifft_kernel and collect_kernel whatever the order the waves
// arrive in: 16 x the largest wave sum (an upper bound, and waves hold similar shares of a row's energy)
// Wave reductions on the VALU's cross-lane paths (DPP inside rows of 16 lanes, four readlanes across the rows): __shfl_*
// goes through ds_bpermute_b32 and needs an address register per step, which ifft_kernel does not have to spare.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <class Op>
__device__ __forceinline__ float row_reduce_f32(float v, Op op) {          // every lane ends with its row's result
    v = op(v, dpp_f32<0xB1>(v));                                 // quad_perm [1,0,3,2]
    v = op(v, dpp_f32<0x4E>(v));                                 // quad_perm [2,3,0,1]
    v = op(v, dpp_f32<0x141>(v));                                // row_half_mirror
    v = op(v, dpp_f32<0x140>(v));                                // row_mirror
    return v;
}
template <class Op>
__device__ __forceinline__ float wave_reduce_f32(float v, Op op) {         // wave-uniform result
    v = row_reduce_f32(v, op);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ float wave_sum_f32(float v) { return wave_reduce_f32(v, [](float a, float b) { return a + b; }); }
__device__ __forceinline__ float wave_min_f32(float v) { return wave_reduce_f32(v, [](float a, float b) { return fminf(a, b); }); }
__device__ __forceinline__ float wave_max_f32(float v) { return wave_reduce_f32(v, [](float a, float b) { return fmaxf(a, b); }); }
// (bit patterns of floats >= 0: the integer order is the float order, and nothing is dropped for being a NaN)
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    return __float_as_uint(wave_reduce_f32(__uint_as_float(v), [](float a, float b) { return __uint_as_float(__float_as_uint(a) > __float_as_uint(b) ? __float_as_uint(a) : __float_as_uint(b)); }));
}

// shared state of one pair's epilogue (ifft_kernel)
struct IfftShared {
    unsigned red_min, red_rs;        // float bits (both >= 0: unsigned order == float order)
    float red_q[FT / 64];            // per wave: energy of its share of the Y row
    int ccnt, unc_any;
};

template <int METHOD>
__device__ __forceinline__ void ifft_one(const IfftArgs& a, const int slot, float* lds, IfftShared& sh, const int tid) {
    unsigned& red_min = sh.red_min; unsigned& red_rs = sh.red_rs; float (&red_q)[FT / 64] = sh.red_q; int& ccnt = sh.ccnt; int& unc_any = sh.unc_any;
    // which pair: by default the workgroup index; with a schedule the pairs that read the same region of the
    // destination stream run back to back on one XCD, so that the prefix-sum lines they share are fetched into that
    // XCD's L2 once instead of once per search
    const int pr = a.order ? a.order[slot] : slot;
    sushi_fft::uint4v yl[4];
    const float q2 = load_y(yl, a.y + (size_t)pr * (FN / 2), tid);     // in flight while the descriptors below arrive
    const sushi_fft::MfmaB mb = dft16_operands(tid);
    const sushi_fft::WTwiddles tw = sushi_fft::load_wtwiddles<1>(tid, twiddles());
    const int lane = tid & 63;
    {
        const float qw = wave_sum_f32(q2);
        if (lane == 0) red_q[tid >> 6] = qw;                           // read after the barriers inside the transform
    }
    const int k = __builtin_amdgcn_readfirstlane(a.pairmap[pr]);       // wave-uniform: everything derived from it is scalar
    const SearchDesc sd = a.searches[k];
    const int i = a.sub_first_pair + pr - sd.first_pair;
    const FftLayout lay = fft_layout(sd.win_start, sd.n_pos, sd.tmpl_len);
    const TemplConsts tc = a.tconst[k];
    // the pair's row of the candidate array starts as all NO_KEY (memset at the start of the run): only what exists is written
    unsigned long long* __restrict__ cout = a.cand + (size_t)pr * FFT_ROW;
    if (METHOD == SUSHI_HIP_METHOD_CCOEFF_NORMED && tc.flat) {
        // a pattern without variance: cv2's result is all ones (refine_kernel answers position 0); nothing to rank
        if (tid == 0) { cout[FFT_CAND + 1] = 0ull; a.pair_lb[pr] = __builtin_inff(); }
        return;
    }
    if (tid == 0) { ccnt = 0; unc_any = 0; red_min = 0x7f800000u; red_rs = 0u; }   // read after the barriers inside the transform
    PairScores ps;
    int plo, phi;
    float zn, zn_c;
    score_pair<METHOD>(a, yl, mb, sd, tc, lay.pair0 + i, lds, tid, tw, ps, plo, phi, zn, zn_c);
    // minimum of the pair and the largest 1/|window| (error bound): 32-bit wave reductions, then one LDS atomic each per
    // wave and ONE barrier -- the workgroup's tail is serial time on a CU that holds two workgroups
    const float wmin = wave_min_f32(ps.best), wrs = wave_max_f32(ps.max_rs);
    if (lane == 0) {
        atomicMin(&red_min, __float_as_uint(wmin));                    // scores are >= 0 or +inf
        atomicMax(&red_rs, __float_as_uint(wrs));
    }
    if (METHOD == SUSHI_HIP_METHOD_CCOEFF_NORMED && __ballot(ps.any_uncertain) != 0ull && lane == 0) unc_any = 1;
    __syncthreads();
    const float lmin_s = __uint_as_float(red_min), rs_max = __uint_as_float(red_rs);
    const bool have_min = lmin_s < __builtin_inff();
    static_assert(FT / 64 == 16, "one row of lanes reads the sixteen wave sums");
    const float qmax = row_reduce_f32(red_q[lane & 15], [](float a, float b) { return fmaxf(a, b); });
    const int mac_passes = (lay.n_seg + MAC_SMAX_LONG - 1) / MAC_SMAX_LONG;
    const float e_model = pair_error_model<METHOD>(zn, zn_c, rs_max, tc, (float)(FT / 64) * qmax, tc.inv_scale, mac_passes);
    const float e_pair = fmaxf(0.5f * a.delta, e_model);
    // positions leave this kernel relative to the search's window: p = (pair's first sample + pos) - win_start
    const int64_t shift = (lay.pair0 + i) * (int64_t)FFT_STEP * FFT_SEG - sd.win_start;
    if (tid == 0) {
        cout[FFT_CAND + 1] = ((unsigned long long)__float_as_uint(e_model) << 32) | __float_as_uint(e_pair);
        // the smallest lower bound among the pair's positions: refine_kernel reads this one float per pair and opens the row
        // only of pairs that can hold the search's extremum (it used to read every row of every pair: 10 KB per search)
        const bool unc = METHOD == SUSHI_HIP_METHOD_CCOEFF_NORMED && unc_any;
        a.pair_lb[pr] = unc ? 0.f : (have_min ? fmaxf(lmin_s - e_pair, 0.f) : __builtin_inff());
        if (a.slb && have_min && !unc) {
            // The pair's lower bound (slb_kernel) must not be above any of its exact scores, and the exact score of its best position
            // is at most lmin_s + e_pair.  Checked on every pair that IS transformed -- among them, per run, one pair per audited
            // search that the bound had excluded (survivor_kernel): where it fails the search is evaluated at every position.
            // (TM_SQDIFF_NORMED scores are clamped at 1, cv2's rule, the bound is not: a pair far louder than the pattern has
            // a bound in the thousands and every score 1)
            const float s = METHOD == SUSHI_HIP_METHOD_CCOEFF_NORMED ? a.slb[pr] : fminf(a.slb[pr], 1.0f), ub = lmin_s + e_pair;
            const unsigned char am = a.audit_mark ? a.audit_mark[pr] : (unsigned char)0;
            const bool audit = (am & 1) != 0;
            if (s > ub * 1.00001f + 1e-7f) {
                a.viol[a.first_search + k] = 1;
                atomicAdd(&a.counters->slb_violations, 1);
            }
            if (audit) {
                atomicAdd(&a.counters->excluded_audited, 1ull);
                if (am & 4) atomicAdd(&a.counters->second_look_audited, 1ull);
                const float ratio = s > 0.f ? s / fmaxf(ub, 1e-30f) : 0.f;
                if (__float_as_uint(ratio) > *(volatile uint32_t*)&a.counters->max_slb_ratio_bits)
                    atomicMax(&a.counters->max_slb_ratio_bits, __float_as_uint(ratio));
            }
        }
    }
    // a position can be the search's minimum only if score - e <= (smallest score + e) of the search; inside the
    // pair that is score <= lmin_s + 2 e (refine_kernel applies the search-wide threshold to the stored lower bounds);
    // uncertain positions (METHOD 1) always can
    const float thr = lmin_s + 2.0f * e_pair;
    const bool mine = METHOD == SUSHI_HIP_METHOD_CCOEFF_NORMED ? ((ps.best <= thr && have_min) || ps.any_uncertain)
                                                               : (ps.best <= thr && have_min);
    if (mine) {                                                 // few lanes (often one wave) get past this
        // slots 0 .. FFT_CAND-1: candidates, written where they are found; whoever draws slot FFT_CAND writes the marker
        // "more candidate positions than slots" (a lower bound of everything unlisted, position 0xffffffff): refine_kernel
        // flags the search if that bound is under the search's threshold
        const float unlisted_lb = (METHOD == SUSHI_HIP_METHOD_CCOEFF_NORMED && unc_any) ? 0.f : fmaxf(lmin_s - e_pair, 0.f);
        int lowest = 0x7fffffff;                                // lowest position of this thread that holds the minimum
#pragma unroll
        for (int q = 0; q < 2 * HPT; ++q) {
            const float sc = ps.scores[q];
            if (sc <= thr && sc < __builtin_inff()) {
                const int pos = tid + FT * (q % HPT) + (q / HPT) * FH;
                if (sc == lmin_s) lowest = pos < lowest ? pos : lowest;
                const int slot = atomicAdd(&ccnt, 1);
                if (slot < FFT_CAND) cout[slot] = make_key(fmaxf(sc - e_pair, 0.f), (unsigned)((int64_t)pos + shift));
                else if (slot == FFT_CAND) cout[FFT_CAND] = make_key(unlisted_lb, 0xffffffffu);
            }
        }
        // the pair's upper bound (its smallest score + e) at the lowest position holding it: ties inside a pair are rare, and
        // atomicMin over the keys settles them by position
        if (lowest != 0x7fffffff)
            atomicMin(a.gkeys + a.first_search + k, make_key(lmin_s + e_pair, (unsigned)((int64_t)lowest + shift)));
    }
    // the pair's audit runs: AUDIT_RUNS x FFT_AUDIT consecutive positions at a pseudo-random place (a hash of the pair index)
    // leave with their plain f32 scores whether or not they are candidates; refine_kernel evaluates AUDIT_RUNS runs per search
    // exactly, from as many different transformed pairs as there are (consecutive positions: their windows are one another's but
    // for a few samples, so a run costs the loads of ONE position)
    {
        constexpr int RUNP = AUDIT_RUNS * FFT_AUDIT;                     // FT % RUNP == 0: one r, one half for the whole stretch
        const unsigned h = ((unsigned)(a.sub_first_pair + pr) * 2654435761u) >> 7;
        const int pos_a = RUNP * (int)(h % (unsigned)(2 * FH / RUNP));
        const int j = tid - pos_a % FT;
        if (j >= 0 && j < RUNP) {
            const int qa = (pos_a / FH) * HPT + (pos_a % FH) / FT;
            float sc = __builtin_inff();
#pragma unroll
            for (int q = 0; q < 2 * HPT; ++q) sc = q == qa ? ps.scores[q] : sc;
            if (sc >= 0.f && sc < __builtin_inff()) cout[FFT_CAND + 2 + j] = make_key(sc, (unsigned)((int64_t)(pos_a + j) + shift));
        }
    }
}

// One workgroup per pair; with `count` (a list whose length only the device knows: the pairs the bound left) a FIXED grid strides
// over the list instead of one workgroup per possible entry -- 346,000 workgroups that found their slot empty were 0.3 ms of a step.
template <int METHOD>
__global__ __launch_bounds__(FT, 8)
void ifft_kernel(IfftArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    __shared__ IfftShared sh;
    if (a.count && (int)blockIdx.x >= *a.count) return;               // a list shorter than the grid
    ifft_one<METHOD>(a, (int)blockIdx.x, lds, sh, (int)threadIdx.x);
}
template <int METHOD>
__global__ __launch_bounds__(FT, 8)
void ifft_list_kernel(IfftArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    __shared__ IfftShared sh;
    const int n = *a.count;
    for (int slot = a.list_first + blockIdx.x; slot < n; slot += gridDim.x) {
        // (everything an iteration needs is loaded inside it, off a thread index the compiler cannot see through: left to hoist
        // the transform's per-thread constants out of the loop it spills them -- collect_kernel's lesson)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        ifft_one<METHOD>(a, slot, lds, sh, tid);
        __syncthreads();                                               // the shared state is consumed before the next pair resets it
    }
}

// ------------------------------------------------------------------------------------------
// Which block pairs need no inverse transform at all.
//
// A position can be the search's arg-min only if its score is not above U, the smallest (f32 score + bound) any pair of the
// search has reported.  bound_kernel gives every pair a LOWER bound of the exact scores of all its positions without scoring
// any of them:
//   * the cross term: after the three passes of the inverse transform that stay inside a wave (no LDS exchange, no barrier),
//     wave n1 holds A_n1[k2], the 1024-point transform of its decimated share of Y, and every output of the full transform is
//     a sum of sixteen of them times unit factors: |y[r]| <= B = sum_n1 max_k2 |A_n1[k2]|.  For a pair that does not hold the
//     match the A are noise and B is a fifth of what the matching pair's is;
//   * the window energies: sum I^2 over [p, p + M) >= S2[(j + M / G) G] - S2[(j + 1) G] for every p of the G-sample stretch j
//     (a table of the prefix at every G = COARSE_G-th sample, built once per stream); wlb = the smallest over the pair's
//     stretches that hold a valid position;
//   * score(p) = (sum T^2 + wU - 2 sum T I) / (|T| sqrt(wU)) increases with wU and decreases with sum T I, so
//         slb = (tU' + wlb - 2 Ymax) / (|T| sqrt(wlb)),   Ymax = B / scale + the cross term's modelled error (pair_error_model)
//     is below every exact score of the pair.
// pilot_kernel then picks, per search, the pair with the smallest slb -- where the match is if there is one --, ifft_kernel
// transforms and scores those (one workgroup per search) and leaves U; survivor_kernel lists the pairs with slb <= U, which are
// transformed as before; all others are done: slb > U means every exact score of the pair is above the exact score of a position
// already found.  (U >= 1 -- nothing matches anywhere, every score clamps to 1 and TIES -- excludes nothing.)
// ------------------------------------------------------------------------------------------
struct BoundArgs {
    const uint2* y;
    const double* dst_stats;
    const SearchDesc* searches;       // the sub-batch's searches
    int sub_first_pair;
    int first_search;
    int64_t dst_len;
    const int* pairmap;
    const TemplConsts* tconst;
    const double* ubase;
    const double* sbase;
    int64_t nb;
    const double* coarse;             // [nc] s2 at every COARSE_G-th sample
    int64_t nc;
    float* slb;                       // [pairs of the sub-batch] out: lower bound of the pair's exact scores (-inf: none)
    float* acc;                       // [pairs of the sub-batch][2] bound_kernel's accumulators: sum of the waves' max |A|, largest wave row energy
    // pilot / survivor stages
    int n_sub;
    int n_pairs;
    int* plist;                       // [searches of the sub-batch] the pair transformed first
    int* slist;                       // [pairs of the sub-batch] the pairs still to transform
    int* scount;                      // [1]
    const int* order;                 // the L2-friendly schedule of all pairs (survivors keep its order)
    const unsigned long long* gkeys;  // [all searches]
    float* pair_lb;
    RunCounters* counters;
    // band-split form (bound_low_kernel: `y` = the low rows; slb_kernel adds the rest of the spectrum from the rows' norms)
    int band;                         // 0: `acc` is over whole rows (bound_kernel); 1: over low rows + norms; 2: norms only (prediction)
    int sub_first_seg;
    const float* tnorm_rest;          // [segments of the sub-batch] pattern spectra: SQUARED norm outside the band
    const float* znorm_rest;          // [3][norm_stride] block spectra: norm outside the band of Z, of its real block at j B, of the one H on
    int64_t norm_stride;
    int* band_votes;                  // [2] prediction: pairs looked at, pairs whose bound leaves room
    unsigned char* audit_mark;        // [pairs of the sub-batch] bit 0 = excluded, transformed all the same (the audit of the exclusion); bit 1 = listed; bit 2 = excluded by the second look
    unsigned audit_seq;               // changes from run to run: which excluded pair of a search is audited
    int audit_every;                  // one search in this many is audited per run (0: none)
    int worst_case;                   // 1: every rounding on the excluded side enters at its WORST CASE (slb_one; the default); 0: round 5's statistical model
    float half_err;                   // what a packed-half transform output may be off by, in units of the largest pass-1 value (bound_low_kernel / bound_kernel)
    const int* list;                  // slb_list_kernel / bound_low_exact_kernel / survivor2_kernel: the pairs the first bound left ...
    const int* list_count;            // ... how many
    int* list2;                       // survivor2_kernel: the pairs the second look left ...
    int* list2_count;                 // ... how many
};

// Stage 1 of the band-split form.  A wave takes a PAIR: the eight groups of its low row one after the other (2 KB each, the next
// one requested before the current one is transformed), each through the three in-wave passes of its half-empty 1024-point
// transform in packed halves (fft_core.hpp "LOW BAND").  Every lane register holds the same output index k for every group, so
// the sum over the groups stays in registers: with y_low[2 r'] = sum_g w^(g r') A_g[r' mod 1024],
//     |y_low[2 r']| <= sum_g |A_g[k]| <= sqrt(8 sum_g |A_g[k]|^2)             (Cauchy-Schwarz over the eight groups)
// -- one v_dot2 per value, no square root, no exchange between waves -- and the largest of that over k bounds the low band's
// transform at its sample points (the sum of the groups' separate maxima is 1.7 x looser).  The halves' rounding (header of the
// packed-half passes: every |A_g| may be 0.2 % + 0.29 x the largest pass-1 value off) goes on top by Minkowski's inequality.
// Persistent, free-running waves; acc[2 pr] = the bound, acc[2 pr + 1] = the low row's energy (plain stores: one wave per pair).
__global__ __launch_bounds__(256, 3)
void bound_low_kernel(BoundArgs a) {
    const int lane = threadIdx.x & 63;
    const int waves = gridDim.x * 4;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_items = (int64_t)a.n_pairs * sushi_fft::LB_GROUPS;
    const sushi_fft::MfmaBl mb = sushi_fft::load_mfma_bl(lane, reinterpret_cast<const sushi_fft::uint2v*>(g_dft16_bl));
    const sushi_fft::HTwiddles tw = sushi_fft::load_htwiddles(lane, twiddles());
    const sushi_fft::uint4v* __restrict__ yh = reinterpret_cast<const sushi_fft::uint4v*>(a.y);
    auto load_item = [&](const int64_t it, sushi_fft::uint4v (&yl)[4]) {
        // item = pair * 8 + group: a pair's groups are consecutive 2 KB pieces of its row (the upper lanes re-read the lower lanes' entries)
#pragma unroll
        for (int u = 0; u < 4; ++u) yl[u] = yh[(size_t)it * (LROWE / sushi_fft::LB_GROUPS) + u * 32 + (lane & 31)];
    };
    if (gw >= a.n_pairs) return;
    sushi_fft::uint4v yl[4], yn[4];
    load_item((int64_t)gw * sushi_fft::LB_GROUPS, yl);
    for (int64_t pr = gw; pr < a.n_pairs; pr += waves) {
        float msum[sushi_fft::PER];
#pragma unroll
        for (int r = 0; r < sushi_fft::PER; ++r) msum[r] = 0.f;
        float q2 = 0.f, d2 = 0.f, dc_re = 0.f, dc_im = 0.f;
        for (int g = 0; g < sushi_fft::LB_GROUPS; ++g) {
            // the next group of this pair, or the first of the wave's next pair (the very last one re-requests itself)
            int64_t nx = pr * sushi_fft::LB_GROUPS + g + 1;
            if (g == sushi_fft::LB_GROUPS - 1) nx = pr + waves < a.n_pairs ? (pr + waves) * sushi_fft::LB_GROUPS : nx - 1;
            nx = nx < n_items ? nx : n_items - 1;
            load_item(nx, yn);
            __builtin_amdgcn_sched_barrier(0);
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            float q = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) q = add_abs2_entry(yl[u], q);
            q2 += lane < 32 ? q : 0.f;
            if (g == 0) {
                // Bin 0 apart: a pattern that is not centred (TM_SQDIFF_NORMED) meets whatever a stretch of the stream sums to
                // beside the stream's mean -- a constant under every output of the pair, often the largest single bin, and
                // with it in, one of the eight groups dwarfs the others (the Cauchy-Schwarz step above is then 2.8 x loose).
                // It is entry 0, sub-position 0 of the row (lb_bin_of); its real / imaginary part is added back SIGNED below.
                const h2 y0 = __builtin_bit_cast(h2, (unsigned)__builtin_amdgcn_readfirstlane((int)yl[0][0]));
                dc_re = (float)y0.x; dc_im = (float)y0.y;
                if ((lane & 31) == 0) yl[0][0] = 0u;
            }
            sushi_fft::h2 v[sushi_fft::PER];
            unsigned in2;
            sushi_fft::fft_wave_half_front_low(yl, v, tw, mb, in2);
            d2 += a.half_err * a.half_err * __uint_as_float(wave_max_u32(in2));   // (half_err x this group's largest pass-1 value)^2
#pragma unroll
            for (int r = 0; r < sushi_fft::PER; ++r) msum[r] = __builtin_amdgcn_fdot2(v[r], v[r], msum[r], false);
#pragma unroll
            for (int u = 0; u < 4; ++u) yl[u] = yn[u];
        }
        unsigned m2 = 0u;
#pragma unroll
        for (int r = 0; r < sushi_fft::PER; ++r) m2 = sushi_fft::h_max_bits(m2, __float_as_uint(msum[r]));   // (sums of squares: >= 0)
        const unsigned wm = wave_max_u32(m2);
        const float qw = wave_sum_f32(q2);
        if (lane == 0) {
            // sqrt(8) (sqrt(max_k sum_g |A_g[k]|^2) (1 + 0.2 %) + sqrt(sum_g (0.29 max |pass-1 value of g|)^2)), the 2^-10 undone:
            // the modulus of the band (bin 0 aside) at its sample points; sqrt(2) more everywhere between them (fft_core.hpp
            // "LOW BAND"); plus bin 0's own part, signed -- the cross term's real parts score the pair's first half, its imaginary
            // parts the second: an UPPER bound of both is what a lower bound of the scores needs
            float bw = 1.4142137f * 2.8284272f * (sqrtf(__uint_as_float(wm)) * 1.002f + sqrtf(d2)) * 1024.0f + fmaxf(dc_re, dc_im);
            if (wm >= 0x7f800000u || !(d2 < __builtin_inff())) bw = __builtin_inff();
            a.acc[2 * (size_t)pr] = bw;
            a.acc[2 * (size_t)pr + 1] = qw * 1.000001f;
        }
    }
}

// Stage 1: the transform part.  An item = (pair, n1): one wave's decimated share of one pair's Y (4 KB), three passes, the
// largest |A_n1[k2]|.  Waves are persistent and free-running -- no workgroup-wide step: every wave walks its own items and
// requests the next item's Y before it transforms the current one, so that the HBM stream never waits for a transform (with
// one workgroup per pair, all sixteen waves loading and then all sixteen transforming, it ran at half the rate).  Neighbouring
// waves take neighbouring items: the sixteen shares of a pair are read at about the same time.  Results are added to the pair's
// accumulators (zeroed by a memset): acc[2 pr] += max |A|, acc[2 pr + 1] = max(row energy of a wave) as float bits.
constexpr int BOUND_THREADS = 256;
__global__ __launch_bounds__(BOUND_THREADS, 4)
void bound_kernel(BoundArgs a) {
    const int lane = threadIdx.x & 63;
    const int waves = gridDim.x * (BOUND_THREADS / 64);
    const int gw = blockIdx.x * (BOUND_THREADS / 64) + (threadIdx.x >> 6);
    const int64_t n_items = (int64_t)a.n_pairs * 16;
    // the transform's per-lane constants do not depend on the wave: loaded (and rounded to halves) once
    const sushi_fft::MfmaBh mb = sushi_fft::load_mfma_bh(lane, reinterpret_cast<const sushi_fft::uint4v*>(g_dft16_bh));
    const sushi_fft::HTwiddles tw = sushi_fft::load_htwiddles(lane, twiddles());
    const sushi_fft::uint4v* __restrict__ yh = reinterpret_cast<const sushi_fft::uint4v*>(a.y);
    auto load_item = [&](const int64_t it, sushi_fft::uint4v (&yl)[4]) {
        const size_t pr = (size_t)(it >> 4);
        const int n1 = (int)(it & 15);
#pragma unroll
        for (int u = 0; u < 4; ++u) yl[u] = yh[pr * (FN / 4) + sushi_fft::wslot_uint4(n1 * 64 + lane, u)];
    };
    int64_t it = gw;
    if (it >= n_items) return;
    sushi_fft::uint4v yl[4], yn[4];
    load_item(it, yl);
    for (; it < n_items; it += waves) {
        const int64_t nx = it + waves < n_items ? it + waves : it;       // (the last item re-requests itself: unconditional loads)
        load_item(nx, yn);
        __builtin_amdgcn_sched_barrier(0);      // (left to itself the scheduler sinks these loads to the end of the loop body: no prefetch at all)
        float q2 = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) q2 = add_abs2_entry(yl[u], q2);
        // the three in-wave passes in packed halves (fft_core.hpp): 2^-10 A_n1[k2], good to two digits -- enough for a bound
        sushi_fft::h2 v[sushi_fft::PER];
        unsigned in2;
        sushi_fft::fft_wave_half_front(yl, v, tw, mb, in2);
        unsigned m2 = 0u;                                                  // largest |value|^2 as float bits (fft_core.hpp h_abs2)
#pragma unroll
        for (int r = 0; r < sushi_fft::PER; ++r) m2 = sushi_fft::h_max_bits(m2, sushi_fft::h_abs2(v[r]));
        const unsigned wm = wave_max_u32(m2), wi = wave_max_u32(in2);
        const float qw = wave_sum_f32(q2);
        if (lane == 0) {
            const size_t pr = (size_t)(it >> 4);
            // the largest |A|: what the halves gave, their rounding (header of the packed-half passes), the 2^-10 undone
            float bw = (sqrtf(__uint_as_float(wm)) * 1.002f + a.half_err * sqrtf(__uint_as_float(wi))) * 1024.0f;
            if (wm >= 0x7f800000u || wi >= 0x7f800000u) bw = __builtin_inff();
            atomicAdd(a.acc + 2 * pr, bw);
            atomicMax(reinterpret_cast<unsigned*>(a.acc + 2 * pr + 1), __float_as_uint(qw));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) yl[u] = yn[u];
    }
}

// Stage 2: one wave per pair puts the bound together (header above): B and the row energy from the accumulators, the lower
// bound of the window energies from the coarse table, the FFT stage's error from the block bases.
// TM_CCOEFF_NORMED (ranked as 1 - cc): cc(p) = sum (T - mean T) I / (|T_c| sqrt(d2(p))) <= Ymax / (|T_c| sqrt(d2lb)) with
//   d2(p) = sum over the window of (I - c)^2  -  (sum over the window of (I - c))^2 / M          (a variance sum: any c)
//         >= Ein(j) - (|Din(j)| + sqrt(2 G (Eout(j) - Ein(j))))^2 / M
// for every p of stretch j: Ein / Din = energy / sum of the centred samples over the G-aligned span inside every such window,
// Eout = the energy over the G-aligned span around every such window; what a window holds beyond the inner span is at most 2 G
// samples of at most Eout - Ein energy (Cauchy-Schwarz).  A flat window anywhere in the pair makes d2lb <= 0: nothing excluded.
template <int METHOD>
__device__ __forceinline__ void slb_one(const BoundArgs& a, const int pr, const int lane) {
    constexpr bool CC = METHOD == SUSHI_HIP_METHOD_CCOEFF_NORMED;
    const int k = __builtin_amdgcn_readfirstlane(a.pairmap[pr]);
    const SearchDesc sd = a.searches[k];
    const FftLayout lay = fft_layout(sd.win_start, sd.n_pos, sd.tmpl_len);
    const int64_t pairI = lay.pair0 + (a.sub_first_pair + pr - sd.first_pair);
    const int64_t qbase = pairI * FFT_STEP * (int64_t)FFT_SEG;
    const int M = sd.tmpl_len;
    const TemplConsts tc = a.tconst[k];
    const int64_t kA = pairI * FFT_STEP;
    const int n_seg = lay.n_seg;
    const int64_t n = a.dst_len;
    const double c = a.dst_stats[1];
    // A wave is a chain of dependent look-ups (pair -> search -> constants -> tables), 354,555 of them at BASELINE configs[2]: what
    // it costs is round trips, not instructions.  So EVERYTHING the wave will read is requested here in one flight, raw, and used
    // only after the table look-ups have been requested too.
    const float acc0 = a.acc[2 * (size_t)pr], acc1 = a.acc[2 * (size_t)pr + 1];
    const int64_t iA = kA < a.nb ? kA : a.nb, iB = kA + n_seg + 2 * FFT_VB < a.nb ? kA + n_seg + 2 * FFT_VB : a.nb;
    const double u0 = a.ubase[iA], u1 = a.ubase[iB], s0 = a.sbase[iA], s1 = a.sbase[iB];
    // band-split form: what the bins outside the band can add to any output of the pair's transform, from the norms of the rows
    // that meet (Cauchy-Schwarz per segment, the triangle inequality over the segments; the stored halves' own rounding and
    // their subnormal floor on top; a pattern of more than MAC_SMAX_LONG segments re-rounds its partial row once per pass).
    // The real parts of the pair's outputs meet the real blocks at 6 I + s, the imaginary parts those H samples on: each from
    // its own block's norm (real_block_rest_norms) -- the larger of the two sums bounds both parts.  That split assumes the
    // pattern rows conjugate-symmetric (spectra of real segments); what their stored halves lack of it (half an ulp per bin
    // and the float32 transform's own asymmetry: < 1.5e-3 of a row's norm) meets the whole |Z|.
    // (unconditional loads from clamped places -- a branch around them would be a second flight --, masked afterwards)
    const int sl = lane < n_seg ? lane : 0;
    const int64_t jl = kA + sl < a.nb ? kA + sl : a.nb;
    float tn0 = a.tnorm_rest[(sd.first_seg - a.sub_first_seg) + sl];
    const float zn0 = a.znorm_rest[jl], an0 = a.znorm_rest[a.norm_stride + jl], bn0 = a.znorm_rest[2 * a.norm_stride + jl];
    // lower bound of the window energies (variance sums): the stretches of COARSE_G positions that hold a valid position
    constexpr int NSB = 2 * FH / COARSE_G;
    static_assert(NSB <= 128 && (FFT_STEP * FFT_SEG) % COARSE_G == 0, "a lane looks up two stretches");
    const int64_t plo = sd.win_start - qbase, phi = (int64_t)sd.n_pos + (sd.win_start - qbase);       // valid: plo <= pos < phi
    const double* __restrict__ c2 = a.coarse;
    const double* __restrict__ c1 = a.coarse + a.nc;
    auto clampi = [&](int64_t x) { return x < a.nc - 1 ? x : a.nc - 1; };
    // (both stretches of a lane: unconditional loads from clamped places, the stretch's validity applied afterwards)
    bool use[2];
    int64_t js[2], je[2], jo0[2], jo1[2];
    double c2e[2], c2s[2], c1e[2], c1s[2], c2o1[2], c2o0[2], c1o1[2], c1o0[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int sb = lane + 64 * t;
        const int64_t p0 = (int64_t)sb * COARSE_G;
        use[t] = sb < NSB && p0 + COARSE_G > plo && p0 < phi;
        const int64_t j = qbase / COARSE_G + sb;
        js[t] = clampi(j + 1); je[t] = clampi(j + M / COARSE_G);
        c2e[t] = c2[je[t]]; c2s[t] = c2[js[t]];
        if (CC) {
            jo0[t] = clampi(j); jo1[t] = clampi(j + M / COARSE_G + 2);
            c1e[t] = c1[je[t]]; c1s[t] = c1[js[t]];
            c2o1[t] = c2[jo1[t]]; c2o0[t] = c2[jo0[t]]; c1o1[t] = c1[jo1[t]]; c1o0[t] = c1[jo0[t]];
        }
    }
    __builtin_amdgcn_sched_barrier(0);          // (nothing that uses a loaded value moves above this line: one flight)
    float wl = __builtin_inff();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (use[t]) {
            double e = 0.0;
            if (je[t] > js[t]) {
                if (CC) {
                    auto len = [&](int64_t lo, int64_t hi) {
                        const int64_t x0 = lo * COARSE_G < n ? lo * COARSE_G : n, x1 = hi * COARSE_G < n ? hi * COARSE_G : n;
                        return (double)(x1 - x0);
                    };
                    const double d_in = (c1e[t] - c1s[t]) - c * len(js[t], je[t]);
                    const double e_in = (c2e[t] - c2s[t]) - 2.0 * c * (c1e[t] - c1s[t]) + c * c * len(js[t], je[t]);
                    const double e_out = (c2o1[t] - c2o0[t]) - 2.0 * c * (c1o1[t] - c1o0[t]) + c * c * len(jo0[t], jo1[t]);
                    const double d_ub = fabs(d_in) + sqrt(2.0 * COARSE_G * fmax(e_out - e_in, 0.0)) * 1.000001 + 1e-6 * (fabs(d_in) + 1.0);
                    e = e_in * 0.999999 - d_ub * d_ub / (double)M - 1e-9 * (c2e[t] - c2s[t]);    // (the prefix table's own rounding)
                } else {
                    e = c2e[t] - c2s[t];
                }
            }
            wl = fminf(wl, fmaxf((float)e * 0.9999995f, 0.f));
        }
    }
    const float wlb = wave_min_f32(wl);
    float b_rest = 0.f;
    if (a.band) {
        const float t0n = lane < n_seg ? sqrtf(tn0) * 1.000002f : 0.f;
        float pz = t0n * zn0, pa = t0n * an0, pb = t0n * bn0;
        for (int s = lane + 64; s < n_seg; s += 64) {               // (patterns of more than 64 segments)
            const int64_t jj = kA + s < a.nb ? kA + s : a.nb;
            const float t = sqrtf(a.tnorm_rest[(sd.first_seg - a.sub_first_seg) + s]) * 1.000002f;
            pz += t * a.znorm_rest[jj]; pa += t * a.znorm_rest[a.norm_stride + jj]; pb += t * a.znorm_rest[2 * a.norm_stride + jj];
        }
        pz = wave_sum_f32(pz); pa = wave_sum_f32(pa); pb = wave_sum_f32(pb);
        const int passes = (n_seg + MAC_SMAX_LONG - 1) / MAC_SMAX_LONG;
        b_rest = (fmaxf(pa, pb) + 1.5e-3f * pz) * tc.mac_scale * (1.0006f + 0.0005f * (float)passes) + 1e-3f;
    }
    if (lane == 0) {
        float B = acc0, qmax = acc1;
        if (a.band) {
            // acc[1] is the low band's energy, and the rest's is at most the square of its sum of moduli -- as one sixteenth of a
            // row's energy, the unit the model below takes
            B = (a.band == 2 ? 0.f : B) + b_rest;                 // (bound_low_kernel's value is complete: sqrt(2) and bin 0 are in it)
            qmax = ((a.band == 2 ? 0.f : qmax) + b_rest * b_rest) * (1.0f / (float)(FT / 64));
        }
        // energy of the samples that enter this pair's transforms, as they are (zn) and centred (zn_c): score_pair's
        const int64_t s_lo = qbase < n ? qbase : n;
        const int64_t s_hi64 = (kA + n_seg + 2 * FFT_VB) * (int64_t)FFT_SEG;
        const int64_t s_hi = s_hi64 < n ? s_hi64 : n;
        const double e2 = (u1 - u0) - 2.0 * c * (s1 - s0) + c * c * (double)(s_hi - s_lo);
        const double zn_c = sqrt(fmax(e2, 0.0)) * 1.0000005, zn = sqrt(fmax(u1 - u0, 0.0)) * 1.0000005;
        const int mac_passes = (n_seg + MAC_SMAX_LONG - 1) / MAC_SMAX_LONG;
        const double sigma_y = sqrt((double)((float)(FT / 64) * qmax) * (7.9472862e-8 * (double)(2 + mac_passes)) + (double)FN * 1.2e-15) *
                               (double)tc.inv_scale;
        // what the exact cross term of any position of this pair can reach: the bound of the transform's outputs (its own
        // float32 rounding included in the factor), the FFT stage's error, the packed halves' modelled error
        const double tn = CC ? (tc.inv_tnorm_c > 0.f ? 1.0 / (double)tc.inv_tnorm_c : 0.0) : (double)tc.tnorm;
        // (an UPPER bound of the signed cross term: the band-split form's may be negative)
        double ymax = (double)B * (double)tc.inv_scale + fabs((double)B) * (double)tc.inv_scale * 2e-5;
        if (a.worst_case) {
            // THE EXCLUDED SIDE IS A PROOF, NOT A MODEL.  B bounds the transform of the halves AS STORED; the exact
            // cross term differs from that by what the stored halves differ from the exact spectra by, every term at its worst case
            // with all of them in phase (triangle inequality over the bins, no independence assumed):
            //   |y[r] - y_stored[r]| <= sum_f |Tt Z - stored product|(f)
            //      <= (2 u_h + 2 e_F + g) sum_s sum_f |Tt_s(f)| |Z_s(f)|  +  (passes) u_h sum_f |Y(f)|       u_h = 2^-11: a half's rounding
            //      <= c sum_s |Tt_s| |Z_s|   (Cauchy-Schwarz over ALL bins, band and rest alike)
            //      =  c sum_s |t_s| |z_s|    (Parseval: segment s of the pattern, the 2 N centred samples block 6 I + s packs)
            //      <= c |T| sqrt(sum_s |z_s|^2) <= c |T| sqrt(8) zn_c
            // (every sample of the pair's span of n_seg + 6 blocks enters at most four block spectra as a first and four as a second
            // half).  e_F = 1e-5 >= 14 (mu + g_4 (sqrt 2 + mu)): the float32 forward transforms' relative error in the 2-norm (Higham,
            // Accuracy and Stability of Numerical Algorithms, Thm 24.2, log2 N = 14 radix-2 levels, twiddles good to 4 u);
            // g <= 2e-5: mac_kernel's float32 sums.  The same terms cover what the NORMS of the stored rows outside the band differ
            // from the exact rows' by, and what the stored pattern rows lack of conjugate symmetry.  The halves' subnormal floor
            // (2^-25 absolute per stored value, N bins a row): of the pattern rows < 1.3e-9 sqrt(n_seg) |T| zn_c (inside c), of the block
            // rows 3.3e-10 sqrt(n_seg) |T| sqrt(E7) (1 / the stream's scale <= 0.01105 sqrt(E7)), of Y itself N 2^-25 of its units.
            const double c_wc = (double)(mac_passes + 2) * 4.8829e-4 * 1.001 + 4e-5;
            const double zn_wc = CC ? fmax(zn, zn_c) : zn_c;
            ymax += c_wc * 2.8284272 * tn * zn_wc
                  + 3.3e-10 * sqrt((double)n_seg) * tn * sqrt(fmax(a.dst_stats[0], 0.0)) + 5e-4 * (double)tc.inv_scale;
        } else {
            ymax += 5.9604645e-8 * (double)FFT_KE * (CC ? fmax(zn, zn_c) : zn_c) * tn + (double)Y_KQ * sigma_y;
        }
        float slb = -__builtin_inff();
        bool room = false;          // (prediction) the bound, with nothing but the norms outside the band in it, keeps 55 % of what a zero cross term would score
        if (CC) {
            if (wlb > 0.f && wlb < __builtin_inff() && !tc.flat && tn > 0.0) {
                slb = (float)(1.0 - ymax / (tn * sqrt((double)wlb)) * 1.000001);
                room = slb > 0.55f;
            }
        } else {
            const double t0 = tc.tU - 2.0 * (double)tc.c_sum_t;
            const double a0 = t0 - 2.0 * ymax;
            if (wlb > 0.f && wlb < __builtin_inff() && a0 < (double)wlb && tc.tU > 0.0) {
                slb = (float)((a0 + (double)wlb) / (sqrt((double)wlb) * (double)tc.tnorm) * 0.999999);
                room = a0 + (double)wlb >= 0.55 * (t0 + (double)wlb);
            }
        }
        a.slb[pr] = slb;
        if (a.band == 2) {
            // prediction: with NOTHING from the low band, does the rest alone leave the bound room to exclude?  (The scores'
            // scale is what a zero cross term gives -- small for streams that sit on a large mean --: the rest may take 45 % of it,
            // the low band's sum and the match's own score need the other half.)
            atomicAdd(a.band_votes, 1);
            if (room) atomicAdd(a.band_votes + 1, 1);
        }
    }
}
template <int METHOD>
__global__ __launch_bounds__(256)
void slb_kernel(BoundArgs a) {
    const int pr = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (pr >= a.n_pairs) return;
    slb_one<METHOD>(a, pr, threadIdx.x & 63);
}
// the same for the pairs of a list (the second look at what the first bound left: bound_low_exact_kernel)
template <int METHOD>
__global__ __launch_bounds__(256)
void slb_list_kernel(BoundArgs a) {
    const int n = *a.list_count;
    if (n > a.n_pairs / 5) return;                      // (nothing was excluded to speak of: no second look either)
    const int waves = gridDim.x * 4;
    for (int slot = blockIdx.x * 4 + (threadIdx.x >> 6); slot < n; slot += waves)
        slb_one<METHOD>(a, __builtin_amdgcn_readfirstlane(a.list[slot]), threadIdx.x & 63);
}

// per search: the pair with the smallest lower bound (the first of them) is transformed first
__global__ __launch_bounds__(64)
void pilot_kernel(BoundArgs a) {
    const int k = blockIdx.x, lane = threadIdx.x;
    const SearchDesc sd = a.searches[k];
    const FftLayout lay = fft_layout(sd.win_start, sd.n_pos, sd.tmpl_len);
    const int p0 = sd.first_pair - a.sub_first_pair;
    unsigned long long best = NO_KEY;
    for (int i = lane; i < lay.n_pairs; i += 64) {
        const float s = a.slb[p0 + i];
        // order-preserving key of a float that may be negative or -inf
        const unsigned b = __float_as_uint(s);
        const unsigned ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
        const unsigned long long key = ((unsigned long long)ord << 32) | (unsigned)i;
        best = key < best ? key : best;
    }
    best = wave_min_u64(best);
    if (lane == 0) {
        a.plist[k] = p0 + (int)(best & 0xffffffffull);
        if (k == 0) *a.scount = 0;
        atomicAdd(&a.counters->pairs_transformed, 1ull);
    }
}

// every pair but the pilots: excluded (its lower bound is above what the search has already found: pair_lb = +inf, what
// refine_kernel and collect_kernel skip by), or listed for ifft_kernel in the order of the L2-friendly schedule
__global__ __launch_bounds__(256)
void survivor_kernel(BoundArgs a) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    int pr = 0;
    if (b < a.n_pairs) {
        pr = a.order[b];
        const int k = a.pairmap[pr];
        bool audit = false;
        if (a.plist[k] != pr) {
            const unsigned long long g = a.gkeys[a.first_search + k];
            const float U = g == NO_KEY ? __builtin_inff() : key_score(g);
            // (a search whose best score is 1 -- no match anywhere, every score clamped to 1 -- ties everywhere: nothing is excluded)
            const bool excluded = U < 0.9999f && a.slb[pr] > U * 1.000001f + 1e-7f;
            if (excluded && a.audit_every > 0 && ((unsigned)(a.first_search + k) + a.audit_seq) % (unsigned)a.audit_every == 0u) {
                // the audit of the exclusion: one hashed pair of the search; if the bound excluded it, it is transformed all the
                // same and ifft_kernel holds its bound to what it really scores
                const SearchDesc sd = a.searches[k];
                const FftLayout lay = fft_layout(sd.win_start, sd.n_pos, sd.tmpl_len);
                const unsigned h = ((unsigned)(a.first_search + k) * 2654435761u + a.audit_seq * 40503u) >> 9;
                audit = (int)(h % (unsigned)lay.n_pairs) == a.sub_first_pair + pr - sd.first_pair;
            }
            if (excluded && !audit) a.pair_lb[pr] = __builtin_inff();
            keep = !excluded || audit;
        }
        if (a.audit_mark) a.audit_mark[pr] = (audit ? 1 : 0) | (keep ? 2 : 0);          // bit 0: audited, bit 1: listed
    }
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && m) {
        base = atomicAdd(a.scount, __popcll(m));
        atomicAdd(&a.counters->pairs_transformed, (unsigned long long)__popcll(m));
    }
    base = __shfl(base, 0, 64);
    if (keep) a.slist[base + __popcll(m & ((1ull << lane) - 1ull))] = pr;
}

// The whole rows of the LISTED pairs, search by search (band-split form, the pairs the bound left): a workgroup = 256 consecutive
// entries of ONE search; the pattern's segment spectra go to registers once and meet the block spectra of every listed pair of
// that search in turn.  mac_list_kernel re-reads the pattern rows for every pair -- 0.6 MB a pair, 8.7 GB a step at BASELINE
// configs[2], what that kernel's time is --; here they are read once per search (1.8 GB), and the block spectra of neighbouring
// searches' pairs, walked at about the same time, meet in the L2.  Patterns of more than MAC_SMAX_LONG segments stay with
// mac_list_kernel (`long_only` there).  Same sums in the same order as mac_kernel (a segment past the pattern's end is a zero entry).
struct MacRowsArgs {
    const uint4* spec;
    int64_t spec_blocks;
    const uint4* tspec;
    uint4* y;
    const SearchDesc* searches;
    const TemplConsts* tconst;
    const unsigned char* mark;        // [pairs of the sub-batch] bit 1: listed
    int n_sub;
    int sub_first_seg;
    int sub_first_pair;
    const int* disable;               // NULL, or a device flag: 1 = the dense multiply-accumulate forms every row instead
};
template <int SMAX>
__device__ __forceinline__ void mac_rows_of_search(const MacRowsArgs& a, const SearchDesc& sd, const FftLayout& lay, const float sy,
                                                   const int e, unsigned long long (*masks)[4], const int z_zero) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const uint4* __restrict__ tsp = a.tspec + (size_t)(sd.first_seg - a.sub_first_seg) * ROWE + e;
    const uint4* __restrict__ zsp = a.spec + e;
    const int p0 = sd.first_pair - a.sub_first_pair;
    sushi_mac::h8 tt[SMAX];
#pragma unroll
    for (int s = 0; s < SMAX; ++s) tt[s] = s < lay.n_seg ? as_h8(tsp[(size_t)s * ROWE]) : sushi_mac::zero_h8();
    const int tid = threadIdx.x;
    for (int base = 0; base < lay.n_pairs; base += MACL_THREADS) {
        // which of these 256 pairs are listed: one flag per thread, a ballot per wave
        __syncthreads();
        const int i = base + tid;
        const bool on = i < lay.n_pairs && (a.mark[p0 + i] & 2);
        const unsigned long long bm = __ballot(on);
        if ((tid & 63) == 0) (*masks)[tid >> 6] = bm;
        __syncthreads();
        for (int w = 0; w < 4; ++w) {
            unsigned long long m = (*masks)[w];
            while (m) {
                const int bit = __builtin_ctzll(m);
                m &= m - 1;
                const int ip = base + 64 * w + bit;
                const long long I = lay.pair0 + ip;
                sushi_mac::acc4 acc = sushi_mac::zero_acc();
#pragma unroll
                for (int s6 = 0; s6 < SMAX; s6 += 6) {
                    sushi_mac::h8 z[6];
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        const long long jj = s6 + t < lay.n_seg ? FFT_STEP * I + s6 + t : (long long)z_zero;     // (past the pattern: the zero block)
                        z[t] = as_h8(zsp[(size_t)(jj < z_zero ? jj : z_zero) * ROWE]);
                    }
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        const sushi_mac::zrow zr = {z[t], sushi_mac::rot_mi(z[t])};
                        sushi_mac::mac4(acc, tt[s6 + t], zr);
                    }
                }
                unsigned o[sushi_mac::BINS];
#pragma unroll
                for (int q = 0; q < sushi_mac::BINS; ++q) {
                    const h2 h = {(_Float16)(acc.re[q] * sy), (_Float16)(acc.im[q] * sy)};
                    o[q] = __builtin_bit_cast(unsigned, h);
                }
                a.y[(size_t)(p0 + ip) * ROWE + e] = uint4{o[0], o[1], o[2], o[3]};
            }
        }
    }
}
template <int LONG>            // 0: patterns of up to MAC_SMAX_SHORT segments; 1: longer ones up to MAC_SMAX_LONG (their registers halve the occupancy)
__global__ __launch_bounds__(MACL_THREADS)
void mac_rows_kernel(MacRowsArgs a) {
    __shared__ unsigned long long masks[4];
    if (a.disable && *a.disable) return;
    const int z_zero = (int)(a.spec_blocks < 0x7fffffff ? a.spec_blocks : 0x7fffffff);
    for (long long it = blockIdx.x; it < (long long)a.n_sub * MACL_PARTS; it += gridDim.x) {
        const int k = (int)(it / MACL_PARTS);
        const int e = (int)(it % MACL_PARTS) * MACL_THREADS + threadIdx.x;
        const SearchDesc sd = a.searches[k];
        const FftLayout lay = fft_layout(sd.win_start, sd.n_pos, sd.tmpl_len);
        const float sy = a.tconst[k].mac_scale;
        if (LONG) {
            if (lay.n_seg > MAC_SMAX_SHORT && lay.n_seg <= MAC_SMAX_LONG) mac_rows_of_search<MAC_SMAX_LONG>(a, sd, lay, sy, e, &masks, z_zero);
        } else {
            if (lay.n_seg <= 6) mac_rows_of_search<6>(a, sd, lay, sy, e, &masks, z_zero);
            else if (lay.n_seg <= 12) mac_rows_of_search<12>(a, sd, lay, sy, e, &masks, z_zero);
            else if (lay.n_seg <= MAC_SMAX_SHORT) mac_rows_of_search<MAC_SMAX_SHORT>(a, sd, lay, sy, e, &masks, z_zero);
        }
        // (longer patterns still: mac_list_kernel, several passes)
    }
}

// A SECOND LOOK at the pairs the band-split bound left (a few per cent of all; most of them belong to the shortest patterns, whose
// match stands least above what chance correlates: profiles/r05/dev/survivor_stats_*.txt).  bound_low_kernel bounds the low band's
// samples by a Cauchy-Schwarz sum over its eight decimated groups; here the N/2 samples themselves are formed -- one 8192-point
// float32 transform of the low row (fft_core.hpp Plan<13>), bin 0 out and signed as there -- and their largest modulus taken: the
// sharpest the sampling bound gets.  slb_list_kernel then redoes the pair's lower bound and survivor2_kernel drops what it now
// excludes, before any whole row is formed.  ~25 ns a listed pair against ~130 for its whole row and transform.
constexpr int BLE_LOGN = 13;
constexpr int BLE_N = sushi_fft::Plan<BLE_LOGN>::N, BLE_T = sushi_fft::Plan<BLE_LOGN>::NT;
static_assert(BLE_N == FN / 2, "the low band sampled at every second position");
__global__ __launch_bounds__(BLE_T)
void bound_low_exact_kernel(BoundArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[sushi_fft::lds_floats<BLE_LOGN>()];
    __shared__ unsigned red[BLE_T / 64];
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const int n = *a.list_count;
    if (n > a.n_pairs / 5) return;
    for (int slot = blockIdx.x; slot < n; slot += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                            // (per-iteration constants stay inside the iteration)
        const sushi_fft::Twiddles tw = sushi_fft::load_twiddles<BLE_LOGN, 1>(tid, twiddles());
        const int pr = a.list[slot];
        const uint32_t* __restrict__ row = reinterpret_cast<const uint32_t*>(a.y) + (size_t)pr * (LROWE * 4);
        cpx v[sushi_fft::PER];
        float dc_re = 0.f, dc_im = 0.f;
#pragma unroll
        for (int r = 0; r < sushi_fft::PER; ++r) {
            const int kin = sushi_fft::in_index<BLE_LOGN>(tid, r);        // index on the N/2 grid: the band is kin < N/8 and kin >= 3N/8
            const bool in_band = kin < FN / 8 || kin >= 3 * FN / 8;
            const int ls = sushi_fft::lslot_of_bin(kin < FN / 8 ? kin : (in_band ? kin + FN / 2 : 0));
            const h2 h = __builtin_bit_cast(h2, in_band ? row[ls] : 0u);
            v[r] = cpx{(float)h.x, (float)h.y};
            if (kin == 0) { dc_re = v[r].x; dc_im = v[r].y; v[r] = cpx{0.f, 0.f}; }
        }
        sushi_fft::fft_split<BLE_LOGN, 1>(v, tid, lds, tw);
        float m2 = 0.f;
#pragma unroll
        for (int r = 0; r < sushi_fft::PER; ++r) m2 = fmaxf(m2, v[r].x * v[r].x + v[r].y * v[r].y);
        const unsigned wm = wave_max_u32(__float_as_uint(m2));
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = wm;
        __syncthreads();
        if (tid == 0) {
            unsigned mm = 0u;
#pragma unroll
            for (int i = 0; i < BLE_T / 64; ++i) mm = red[i] > mm ? red[i] : mm;
            // sqrt(2) x the largest sample (+ the float32 transform's own error: 1e-5 of the inputs' sum of moduli, itself at most
            // sqrt(bins x energy), acc[1]) + bin 0's own part, signed
            const float q = a.acc[2 * (size_t)pr + 1];
            float bw = 1.4142137f * (sqrtf(__uint_as_float(mm)) * 1.00001f + 1e-5f * sqrtf(4096.0f * q)) + fmaxf(dc_re, dc_im);
            if (mm >= 0x7f800000u) bw = __builtin_inff();
            // (never above the first look's: both are bounds)
            a.acc[2 * (size_t)pr] = fminf(a.acc[2 * (size_t)pr], bw);
        }
        __syncthreads();
    }
}
// ... and what the second look leaves: the list again, without the pairs whose new bound excludes them
__global__ __launch_bounds__(256)
void survivor2_kernel(BoundArgs a) {
    const int n = *a.list_count;
    const bool second_look = n <= a.n_pairs / 5;
    for (int base = blockIdx.x * 256; base < n; base += gridDim.x * 256) {
        const int slot = base + threadIdx.x;
        bool keep = false;
        int pr = 0;
        if (slot < n) {
            pr = a.list[slot];
            keep = true;
            if (second_look && !(a.audit_mark[pr] & 1)) {
                const int k = a.pairmap[pr];
                const unsigned long long g = a.gkeys[a.first_search + k];
                const float U = g == NO_KEY ? __builtin_inff() : key_score(g);
                if (U < 0.9999f && a.slb[pr] > U * 1.000001f + 1e-7f) {
                    // the audit of THIS bound: a hashed sample of the pairs it excludes -- other ones every run -- stays listed, is
                    // transformed all the same, and ifft_kernel holds the pair's (second) lower bound to what it really scores
                    const unsigned h = ((unsigned)(a.sub_first_pair + pr) * 2654435761u + a.audit_seq * 40503u) >> 11;
                    if (a.audit_every > 0 && h % (16u * (unsigned)a.audit_every) == 0u) {
                        a.audit_mark[pr] = 1 | 2 | 4;                   // audited, listed, by the second look
                    } else {
                        keep = false;
                        a.pair_lb[pr] = __builtin_inff();
                        a.audit_mark[pr] = 0;
                    }
                }
            }
        }
        const unsigned long long m = __ballot(keep), act = __ballot(slot < n);
        const int lane = threadIdx.x & 63;
        const int dropped = __popcll(act) - __popcll(m);
        int b0 = 0;
        if (lane == 0 && m) b0 = atomicAdd(a.list2_count, __popcll(m));
        if (lane == 0 && dropped > 0)                          // (they were counted as transformed when they were listed)
            atomicAdd(&a.counters->pairs_transformed, (unsigned long long)0 - (unsigned long long)dropped);
        b0 = __shfl(b0, 0, 64);
        if (keep) a.list2[b0 + __popcll(m & ((1ull << lane) - 1ull))] = pr;
    }
}

// Band-split form, after survivor_kernel: forming whole rows pair by pair (mac_list_kernel, ~110 ns a pair) beats the dense
// multiply-accumulate over ALL pairs only while few are listed -- a batch whose searches find no match excludes nothing.  dense[0]
// = 1 hands the rows to mac_kernel's whole-row form instead (both launches are queued; the one not needed leaves at once).
__global__ void dense_mode_kernel(const int* scount, int n_pairs, int* dense) { *dense = *scount > n_pairs / 5 ? 1 : 0; }

// Collection pass over the flagged searches of a sub-batch: the same transforms and scores again (bit for bit), now
// against the search's final threshold; every candidate goes to its tile's list, tiles with many candidates (or
// when the buffer is full) are marked dense.  A fixed grid strides over the (search, pair) items refine_kernel listed.
constexpr int SPARSE_MAX = SPARSE_TILE_MAX; // candidates per tile up to which they are listed; beyond: every position
constexpr int COLLECT_GRID = 1024;         // collect_kernel's workgroups: two per CU, twice over

template <int METHOD>
__global__ __launch_bounds__(FT, 8)
void collect_kernel(IfftArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    __shared__ int tcnt[TILES_PER_PAIR], toff[TILES_PER_PAIR], tfill[TILES_PER_PAIR];
    const int n_items = *a.n_citems;
    if (n_items == 0) return;
    // refine_kernel listed, for every search it flagged, the pairs that can hold a candidate (all pairs for a search that goes
    // to every position): a fixed grid strides over that list -- every workgroup has work while there is any
    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
        {
            // everything an iteration needs is loaded inside it, off a thread index the compiler cannot see through: left to
            // hoist the transform's per-thread constants and addresses out of the loop it spilled hundreds of bytes of them
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const sushi_fft::WTwiddles tw = sushi_fft::load_wtwiddles<1>(tid, twiddles());
            const sushi_fft::MfmaB mb = dft16_operands(tid);
            const int lane = tid & 63;
            const int pr = __builtin_amdgcn_readfirstlane(a.citems[it]);
            const int k = __builtin_amdgcn_readfirstlane(a.pairmap[pr]);
            const int s_idx = a.first_search + k;                              // global search index
            const SearchDesc sd = a.searches[k];
            const FftLayout lay = fft_layout(sd.win_start, sd.n_pos, sd.tmpl_len);
            const TemplConsts tc = a.tconst[k];
            const bool everything = a.flags[s_idx] == 2;
            // smallest (score + bound) of the search; none (TM_CCOEFF_NORMED: every window uncertain): everything is a candidate
            const float U = a.gkeys[s_idx] == NO_KEY ? 4.0f : key_score(a.gkeys[s_idx]);
            const int i = a.sub_first_pair + pr - sd.first_pair;
            const int64_t qbase = (lay.pair0 + i) * (int64_t)FFT_STEP * FFT_SEG;
            const int64_t rel0 = qbase - sd.win_start;                     // position (relative to the window) of pos 0
            __syncthreads();                                               // previous item's shared state is consumed
            if (tid < TILES_PER_PAIR) { tcnt[tid] = 0; tfill[tid] = 0; toff[tid] = -1; }
            unsigned candmask = 0;                                         // bit q: scores[q] is a candidate
            int plo = 0, phi = 0;
            if (!everything) {
                PairScores ps;
                float zn, zn_c;
                sushi_fft::uint4v yl[4];
                (void)load_y(yl, a.y + (size_t)pr * (FN / 2), tid);
                score_pair<METHOD>(a, yl, mb, sd, tc, lay.pair0 + i, lds, tid, tw, ps, plo, phi, zn, zn_c);
                // the bound the pair was ranked with, as ifft_kernel stored it (the same scores again, bit for bit)
                const float e_pair = __uint_as_float((unsigned)(a.cand[(size_t)pr * FFT_ROW + FFT_CAND + 1] & 0xffffffffull));
#pragma unroll
                for (int q = 0; q < 2 * HPT; ++q) {
                    const bool c = fmaxf(ps.scores[q] - e_pair, 0.f) <= U;              // invalid positions hold +inf, uncertain ones -1
                    candmask |= c ? (1u << q) : 0u;
                    // all threads of the workgroup look at the same tile for a given q
                    const int tile = (FT * (q % HPT)) / TILE + (q / HPT) * (FH / TILE);
                    const unsigned long long b = __ballot(c);
                    if (lane == 0 && b) atomicAdd(&tcnt[tile], __popcll(b));
                }
            } else {
                const int64_t plo64 = sd.win_start - qbase;
                plo = (int)(plo64 < 0 ? 0 : plo64);
                const int64_t phi64 = (int64_t)sd.n_pos + (sd.win_start - qbase);
                phi = (int)(phi64 < 2 * FH ? phi64 : 2 * FH);
                __syncthreads();
                if (tid < TILES_PER_PAIR) {                                 // every tile that holds a valid position
                    const int t0 = tid * TILE;
                    if (t0 < phi && t0 + TILE > plo) tcnt[tid] = TILE;
                }
            }
            __syncthreads();
            if (tid == 0) {
                // One thread hands out this pair's tiles: ONE addition to each of the run's counters per workgroup (a tile at a time,
                // from 24 threads of every workgroup, the same-address atomics -- and a compare-and-swap loop among them -- were
                // most of this kernel's time).  The candidate buffer's counter never runs far past its capacity: a workgroup that
                // sees it full does not add to it (its tiles are evaluated at every position instead).
                // (a listed tile goes to exact_tiles_kernel in entries of at most SPARSE_UNIT candidates: what an entry costs is
                // bounded, and a workgroup's threads are all busy with it)
                int total = 0, nt = 0, n_sparse = 0, extra = 0;
                for (int t = 0; t < TILES_PER_PAIR; ++t) {
                    const int cnt = tcnt[t];
                    if (cnt > 0) { ++nt; if (cnt <= SPARSE_MAX) { total += cnt; ++n_sparse; extra += (cnt - 1) / SPARSE_UNIT; } }
                }
                if (nt > 0) {
                    int base = -1;
                    if (total > 0 && *(volatile int*)&a.counters->n_cand + total <= a.cand_cap) {
                        const int o = atomicAdd(&a.counters->n_cand, total);
                        if (o + total <= a.cand_cap) base = o;
                    }
                    // capacity: every tile of every pair + one entry per SPARSE_UNIT candidates of the candidate buffer
                    int slot = atomicAdd(&a.counters->n_tiles, nt + (base >= 0 ? extra : 0));
                    for (int t = 0; t < TILES_PER_PAIR; ++t) {
                        const int cnt = tcnt[t];
                        if (cnt <= 0) continue;
                        int off = -1;
                        if (cnt <= SPARSE_MAX && base >= 0) { off = base; base += cnt; }
                        toff[t] = off;
                        TileDesc td;
                        td.search = s_idx;
                        td.p0 = (int)(rel0 + (int64_t)t * TILE);
                        if (off >= 0) {
                            for (int u = 0; u < cnt; u += SPARSE_UNIT) {
                                td.off = off + u;
                                td.cnt = cnt - u < SPARSE_UNIT ? cnt - u : SPARSE_UNIT;
                                a.tiles[slot++] = td;
                            }
                        } else {
                            td.off = -1;
                            td.cnt = -1;
                            a.tiles[slot++] = td;
                        }
                    }
                    const int listed = base >= 0 || total == 0 ? n_sparse : 0;  // (all of the pair's sparse tiles are listed, or none)
                    if (listed > 0) {
                        atomicAdd(&a.counters->tiles_sparse, (unsigned long long)listed);
                        atomicAdd(&a.counters->candidates, (unsigned long long)total);
                    }
                    if (nt - listed > 0) atomicAdd(&a.counters->tiles_dense, (unsigned long long)(nt - listed));
                }
            }
            __syncthreads();
            if (candmask) {
#pragma unroll
                for (int q = 0; q < 2 * HPT; ++q) {
                    if (candmask & (1u << q)) {
                        const int tile = (FT * (q % HPT)) / TILE + (q / HPT) * (FH / TILE);
                        const int off = toff[tile];
                        if (off >= 0) {
                            const int pos = tid + FT * (q % HPT) + (q / HPT) * FH;
                            const int slot = atomicAdd(&tfill[tile], 1);
                            a.candbuf[off + slot] = (int32_t)(rel0 + pos);
                        }
                    }
                }
            }
        }
    }
}

inline int launch_ok() { return hipGetLastError() == hipSuccess ? SUSHI_HIP_OK : SUSHI_HIP_ELAUNCH; }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

inline int64_t cand_capacity(int64_t pairs) {
    const int64_t want = pairs * 64;
    return want < (1 << 16) ? (1 << 16) : (want > (1 << 24) ? (1 << 24) : want);
}

// bytes of workspace for one sub-batch of `pairs` block pairs, `segs` pattern segments and `searches` searches
struct WsLayout { size_t tspec, y, cand, pair_lb, pairmap, tconst, tiles, candbuf, dummy, slb, acc, plist, slist, scount, citems,
                  tspec_low, ylow, tnorm_rest, audit_mark, slist2, total; };
inline WsLayout ws_layout(int64_t pairs, int64_t segs, int64_t searches) {
    WsLayout w;
    size_t o = 0;
    w.tspec = o; o += align_up((size_t)segs * ROW_BYTES, 256);                   // packed halves: 4 bytes per bin
    w.y = o; o += align_up((size_t)pairs * FN * 2 * sizeof(uint16_t), 256);      // packed halves: 4 bytes per bin
    w.cand = o; o += align_up((size_t)pairs * FFT_ROW * sizeof(unsigned long long), 256);
    w.pair_lb = o; o += align_up((size_t)pairs * sizeof(float), 256);
    w.pairmap = o; o += align_up((size_t)pairs * sizeof(int), 256);
    w.tconst = o; o += align_up((size_t)searches * sizeof(TemplConsts), 256);
    w.tiles = o; o += align_up(((size_t)pairs * TILES_PER_PAIR + (size_t)cand_capacity(pairs) / SPARSE_UNIT + 1) * sizeof(TileDesc), 256);
    w.candbuf = o; o += align_up((size_t)cand_capacity(pairs) * sizeof(int32_t), 256);
    w.dummy = o; o += align_up((size_t)MAC_DUMMY_LINES * MAC_THREADS * sizeof(uint4), 256);
    w.slb = o; o += align_up((size_t)pairs * sizeof(float), 256);
    w.acc = o; o += align_up((size_t)pairs * 2 * sizeof(float), 256);
    w.plist = o; o += align_up((size_t)searches * sizeof(int), 256);
    w.slist = o; o += align_up((size_t)pairs * sizeof(int), 256);
    w.scount = o; o += 256;                                                      // [0] survivors, [1] collect items, [2..3] band prediction votes, [4] dense whole rows, [5] pairs left after the second look
    w.citems = o; o += align_up((size_t)pairs * sizeof(int), 256);
    // band-split exclusion: low-band rows of the pattern spectra and of the products, the pattern rows' norms outside the band
    w.tspec_low = o; o += align_up((size_t)segs * LROW_BYTES, 256);
    w.ylow = o; o += align_up((size_t)pairs * LROW_BYTES, 256);
    w.tnorm_rest = o; o += align_up((size_t)segs * sizeof(float), 256);
    w.audit_mark = o; o += align_up((size_t)pairs, 256);
    w.slist2 = o; o += align_up((size_t)pairs * sizeof(int), 256);
    w.total = o;
    return w;
}

// ---- optional per-stage timing (sushi_hip_profile_begin / _end): HIP events on the launch streams ----
struct ProfSpan { hipEvent_t t0, t1; int stage; };
struct ProfCall { std::vector<ProfSpan> spans; };
bool g_prof_on = false;
std::vector<ProfCall> g_prof;

inline hipEvent_t prof_begin(ProfCall* pc, hipStream_t st) {
    if (!pc) return nullptr;
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    (void)hipEventRecord(e, st);
    return e;
}
inline void prof_end(ProfCall* pc, hipEvent_t t0, int stage, hipStream_t st) {
    if (!pc || !t0) return;
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, st);
    pc->spans.push_back(ProfSpan{t0, e, stage});
}

// ---- host-side plan of a batch: sub-batches that fit the workspace, the inverse-transform schedule, the
// multiply-accumulate work items ------------------------------------------------------------------------
struct SubBatch {
    int a0, b0;                         // searches [a0, b0)
    int64_t pairs, segs;
    int first_pair, first_seg;
    int item_first[2];                  // [mac_kernel, mac_long_kernel]: into the item array (items of 1 + MAC_SPW ints)
    int item_count[2];
    int chunk_group[2];                 // bin chunks an XCD works on at a time
    int long_patterns;                  // searches whose pattern has more than MAC_SMAX_LONG segments
};

struct Plan {
    std::vector<SubBatch> subs;
    std::vector<int32_t> order;         // [total pairs] per sub-batch: workgroup -> pair
    std::vector<int32_t> items;         // [total items][1 + MAC_SPW]: class, then search indices inside the sub-batch or -1
    int64_t pairs = 0, segs = 0;
    size_t ws_bytes = 0;                // workspace the plan was cut for
};

// workspace of the most demanding single search / of the whole batch as one sub-batch
void ws_extremes(const std::vector<SearchDesc>& s, size_t* need_one, size_t* need_all) {
    size_t one = 0;
    int64_t pairs = 0, segs = 0;
    for (const SearchDesc& d : s) {
        const FftLayout l = fft_layout(d.win_start, d.n_pos, d.tmpl_len);
        one = std::max(one, ws_layout(l.n_pairs, l.n_seg, 1).total);
        pairs += l.n_pairs; segs += l.n_seg;
    }
    *need_one = one;
    *need_all = ws_layout(pairs, segs, (int64_t)s.size()).total;
}

int build_plan(const std::vector<SearchDesc>& s, size_t ws_bytes, Plan& plan) {
    const int n = (int)s.size();
    plan.ws_bytes = ws_bytes;
    for (int a0 = 0; a0 < n;) {
        int b0 = a0;
        int64_t pairs = 0, segs = 0;
        while (b0 < n) {
            const FftLayout l = fft_layout(s[b0].win_start, s[b0].n_pos, s[b0].tmpl_len);
            if (ws_layout(pairs + l.n_pairs, segs + l.n_seg, b0 - a0 + 1).total > ws_bytes) break;
            pairs += l.n_pairs; segs += l.n_seg; ++b0;
        }
        if (b0 == a0) return SUSHI_HIP_ENOSPACE;
        SubBatch sb;
        sb.a0 = a0; sb.b0 = b0; sb.pairs = pairs; sb.segs = segs;
        sb.long_patterns = 0;
        for (int k = a0; k < b0; ++k)
            if ((s[k].tmpl_len + FFT_SEG - 1) / FFT_SEG > mac_class_smax(MAC_CLASSES - 1)) ++sb.long_patterns;
        sb.first_pair = s[a0].first_pair; sb.first_seg = s[a0].first_seg;
        // every pair of the sub-batch, keyed by the region of the destination stream it scores (its absolute pair
        // index); workgroup b runs on XCD b % 8 (observed; speed only)
        struct Item { int64_t region; int pair; };
        std::vector<Item> lists[8];
        int pair = 0;
        for (int k = a0; k < b0; ++k) {
            const FftLayout l = fft_layout(s[k].win_start, s[k].n_pos, s[k].tmpl_len);
            for (int i = 0; i < l.n_pairs; ++i, ++pair) {
                const int64_t region = l.pair0 + i;
                lists[region & 7].push_back(Item{region, pair});
            }
        }
        size_t head[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int x = 0; x < 8; ++x)
            std::stable_sort(lists[x].begin(), lists[x].end(), [](const Item& p, const Item& q) { return p.region < q.region; });
        for (int64_t b = 0; b < pairs; ++b) {
            int x = (int)(b & 7);
            if (head[x] >= lists[x].size()) {               // that XCD's queue is exhausted: take from the fullest one
                size_t best = 0;
                for (int y = 0; y < 8; ++y) {
                    const size_t left = lists[y].size() - head[y];
                    if (left > best) { best = left; x = y; }
                }
            }
            plan.order.push_back(lists[x][head[x]++].pair);
        }
        // multiply-accumulate items: MAC_SPW searches of one segment-count class whose windows START next to each other
        // (a wave walks the union of its searches' block ranges with every lane computing, so an item costs
        // union x class size whatever its members need: sorted by window start, eight neighbours of a class differ by as
        // little as that class allows -- in request order the windows of neighbouring events can be a whole window
        // apart, which cost 18 % more rows at BASELINE configs[2]); the items themselves in stream order whatever
        // their class
        // One item list per kernel: classes 0 .. 2 (mac_kernel), classes 3 .. 5 (mac_long_kernel).
        for (int kern = 0; kern < 2; ++kern) {
            struct Item { int64_t first; int cls; std::vector<int> members; };
            std::vector<Item> its;
            std::vector<int> of_class[MAC_CLASSES];
            int n_members = 0;
            double win_blocks = 0.0;
            int64_t ws_lo = INT64_MAX, ws_hi = INT64_MIN;
            for (int k = a0; k < b0; ++k) {
                const FftLayout l = fft_layout(s[k].win_start, s[k].n_pos, s[k].tmpl_len);
                const int c = mac_class(l.n_seg);
                if ((c >= MAC_SHORT_CLASSES) != (kern == 1)) continue;
                of_class[c].push_back(k);
                ++n_members; win_blocks += (double)s[k].n_pos / FFT_SEG;
                ws_lo = std::min(ws_lo, s[k].win_start); ws_hi = std::max(ws_hi, s[k].win_start);
            }
            for (int c = 0; c < MAC_CLASSES; ++c) {
                std::vector<int>& v = of_class[c];
                std::stable_sort(v.begin(), v.end(), [&](int p, int q) { return s[p].win_start < s[q].win_start; });
                for (size_t i = 0; i < v.size(); i += MAC_SPW) {
                    Item it{s[v[i]].win_start, c, {}};
                    for (size_t g = i; g < v.size() && g < i + MAC_SPW; ++g) it.members.push_back(v[g] - a0);
                    its.push_back(it);
                }
            }
            std::stable_sort(its.begin(), its.end(), [](const Item& p, const Item& q) { return p.first < q.first; });
            sb.item_first[kern] = (int)(plan.items.size() / (1 + MAC_SPW));
            for (const Item& it : its) {
                plan.items.push_back(it.cls);
                for (int g = 0; g < MAC_SPW; ++g) plan.items.push_back(g < (int)it.members.size() ? it.members[g] : -1);
            }
            sb.item_count[kern] = (int)its.size();
            // how many items overlap a row of block spectra: window length / spacing of the items' windows.  An XCD keeps
            // 32 CUs x (3 | 2) workgroups in flight; with chunk_group = that / overlap the items in flight per chunk are
            // the ones that share rows (mac_kernel's comment).
            int cg = 1;
            if (n_members > 0) {
                win_blocks /= n_members;
                const double span_blocks = (double)(ws_hi - ws_lo) / FFT_SEG;
                const double spacing = its.size() > 1 ? std::max(span_blocks / (double)(its.size() - 1), 1e-3) : win_blocks;
                const double overlap = std::max(win_blocks / spacing, 1.0);
                const double in_flight = kern == 0 ? 96.0 : 64.0;
                while (cg < MAC_CHUNKS / 8 && in_flight / overlap >= 1.5 * cg) cg *= 2;
            }
            sb.chunk_group[kern] = cg;
        }
        plan.subs.push_back(sb);
        plan.pairs += pairs; plan.segs += segs;
        a0 = b0;
    }
    return SUSHI_HIP_OK;
}

// largest tile variant whose grid still gives the chip (256 CUs x 4 SIMDs) a few waves per SIMD
int choose_direct_variant(const SushiHipRequest* req, int n) {
    const int waves[3] = {1, 4, 4};
    int best = 0;
    for (int v = 0; v < direct_variant_count() && v < 3; ++v) {
        const int tp = direct_variant_tile(v);
        int64_t nt = 0;
        for (int k = 0; k < n; ++k) nt += (req[k].n_pos + tp - 1) / tp;
        if (nt * waves[v] >= 4096) best = v;
    }
    return best;
}

// device-memory layout of a batch
struct BatchLayout { size_t desc, keys, flags, viol, flag_list, sub_flagged, counters, order, items, ws, total; };

BatchLayout batch_layout(int n, int path, int64_t total_pairs, size_t n_item_ints, size_t ws_bytes) {
    BatchLayout b;
    size_t o = 0;
    b.desc = o; o += align_up((size_t)n * sizeof(SearchDesc), 256);
    b.keys = o; o += align_up((size_t)2 * n * sizeof(unsigned long long), 256);
    b.flags = o; o += align_up((size_t)n * sizeof(int), 256);
    b.viol = o; o += align_up((size_t)n * sizeof(int), 256);
    b.flag_list = o; o += align_up((size_t)n * sizeof(int), 256);
    b.sub_flagged = o; o += 256;
    b.counters = o; o += align_up(sizeof(RunCounters), 256);
    b.order = o; o += path == SUSHI_HIP_PATH_FFT ? align_up((size_t)total_pairs * sizeof(int32_t), 256) : 0;
    b.items = o; o += path == SUSHI_HIP_PATH_FFT ? align_up(n_item_ints * sizeof(int32_t), 256) : 0;
    b.ws = o; o += path == SUSHI_HIP_PATH_FFT ? align_up(ws_bytes, 256) : 0;
    b.total = o;
    return b;
}

// requests -> descriptors with their running sums; EINVAL for a malformed request
int make_descs(const SushiHipRequest* req, int n, int variant, std::vector<SearchDesc>& out, int64_t* n_tiles) {
    out.resize(n);
    const int tp = direct_variant_tile(variant);
    int64_t tiles = 0, pairs = 0, segs = 0;
    for (int k = 0; k < n; ++k) {
        const SushiHipRequest& r = req[k];
        if (r.tmpl_len < 1 || r.n_pos < 1 || r.win_start < 0 || r.tmpl_off < 0) return SUSHI_HIP_EINVAL;
        if (r.n_pos > 0x7fffffff - 65536 || r.tmpl_len > 0x7fffffff - 65536) return SUSHI_HIP_EINVAL;
        SearchDesc d;
        d.tmpl_off = r.tmpl_off; d.win_start = r.win_start; d.tmpl_len = r.tmpl_len; d.n_pos = r.n_pos;
        if (tiles > 0x7fffffff || pairs > 0x7fffffff || segs > 0x7fffffff) return SUSHI_HIP_EINVAL;
        d.first_tile = (int32_t)tiles; d.first_pair = (int32_t)pairs; d.first_seg = (int32_t)segs; d.reserved = 0;
        const FftLayout l = fft_layout(r.win_start, r.n_pos, r.tmpl_len);
        tiles += (r.n_pos + tp - 1) / tp;
        pairs += l.n_pairs;
        segs += l.n_seg;
        out[k] = d;
    }
    if (tiles > 0x7fffffff || pairs > 0x7fffffff || segs > 0x7fffffff) return SUSHI_HIP_EINVAL;
    *n_tiles = tiles;
    return SUSHI_HIP_OK;
}

size_t resolve_ws(const std::vector<SearchDesc>& descs, size_t cap) {
    size_t need_one, need_all;
    ws_extremes(descs, &need_one, &need_all);
    if (cap == 0) return need_all;
    return std::max(need_one, std::min(need_all, cap));
}

}  // namespace

// The opaque batch handle of the C ABI.
struct SushiHipBatch {
    const SushiHipStream* dst;
    const SushiHipStream* src;
    int n, path, variant, method, exclusion;
    int band;                           // the exclusion's form in AUTO / ALWAYS: -1 not decided yet, 0 whole rows (bound_kernel), 1 band-split
    int band_decided_method;            // ... which was decided for this method (the pattern spectra differ)
    int band_votes[2];                  // what the decision was taken from: pairs looked at, pairs whose bound leaves room
    unsigned run_seq;                   // runs so far: rotates which excluded pairs are audited
    int audit_every;                    // one search in this many has one excluded pair transformed as a check, per run
    int bound_model;                    // SUSHI_HIP_BOUND_WORST_CASE (default) / _STATISTICAL: how the excluded side's roundings enter slb
    int last_band;                      // form of the exclusion the last run's last sub-batch used (-1: none)
    // AUTO learns from its own runs: a batch whose exclusion excluded next to nothing (searches without a match anywhere) runs
    // without it from then on, looking again every 64th run.  The last run's counts come back through 16 bytes of pinned host memory
    // behind an event that is only ever QUERIED: a run never waits for an earlier one.
    unsigned long long* host_stats;     // [2] pairs transformed, excluded pairs audited
    hipEvent_t stats_ready;
    bool stats_pending;
    int suspended;                      // 1: the exclusion is left out (AUTO)
    unsigned suspended_at;              // run_seq of the run that showed it
    int last_suspended;                 // whether the last run was one of those
    int32_t* packed_out;                // NULL, or where every run ALSO leaves its results as 8-byte (index, score bits) records
    int64_t n_tiles;
    int64_t direct_pairs;               // pairs of the last run's sub-batches that were transformed without the exclusion
    std::vector<SearchDesc> descs;
    Plan plan;
    BatchLayout lay;
    char* mem;
    double flops, algorithmic_bytes;
    hipStream_t last_stream;
    bool ran;
    hipEvent_t uploaded;                // recorded on the create-time stream behind the descriptor / plan uploads
    ~SushiHipBatch() {
        if (uploaded) (void)hipEventDestroy(uploaded);
        if (stats_pending && stats_ready) (void)hipEventSynchronize(stats_ready);       // the last run's counts may still be on their way
        if (stats_ready) (void)hipEventDestroy(stats_ready);
        if (host_stats) (void)hipHostFree(host_stats);
    }
};

extern "C" {

int sushi_hip_fft_size(void) { return FN; }

int sushi_hip_fft_block(void) { return FFT_SEG; }

int sushi_hip_fft_slot_of_bin(int bin) { return (bin < 0 || bin >= FN) ? -1 : sushi_fft::mslot_of_bin(bin); }

int sushi_hip_fft_low_slot_of_bin(int bin) { return sushi_fft::lslot_of_bin(bin); }

size_t sushi_hip_stream_spectra_bytes(int64_t n) {
    if (n <= 0) return 0;
    const size_t rows = (size_t)((n + FFT_SEG - 1) / FFT_SEG + 1);
    return rows * ROW_BYTES + rows * LROW_BYTES + 3 * align_up(rows * sizeof(float), 256);   // norms outside the band: of Z, of its two real blocks
}

int sushi_hip_fft_layout(int64_t win_start, int32_t n_pos, int32_t tmpl_len, int32_t* n_pairs, int32_t* n_seg) {
    if (win_start < 0 || n_pos < 1 || tmpl_len < 1 || !n_pairs || !n_seg) return SUSHI_HIP_EINVAL;
    const FftLayout l = fft_layout(win_start, n_pos, tmpl_len);
    *n_pairs = l.n_pairs;
    *n_seg = l.n_seg;
    return SUSHI_HIP_OK;
}

int sushi_hip_stream_add_spectra(SushiHipStream* s, void* mem_dev, size_t mem_bytes, void* hip_stream) {
    if (!s || !mem_dev) return SUSHI_HIP_EINVAL;
    if ((uintptr_t)mem_dev & 255) return SUSHI_HIP_EALIGN;
    const size_t need = sushi_hip_stream_spectra_bytes(s->n);
    if (mem_bytes < need) return SUSHI_HIP_ENOSPACE;
    if (s->blocks >= 0x7fffffff) return SUSHI_HIP_EINVAL;
    // one block more than the stream has: its samples are all past the end, so its spectrum is zero
    const size_t rows = (size_t)s->blocks + 1;
    uint4* low = (uint4*)((char*)mem_dev + rows * ROW_BYTES);
    float* zn = (float*)((char*)mem_dev + rows * ROW_BYTES + rows * LROW_BYTES);
    if (s->dtype == SUSHI_HIP_F32)
        hipLaunchKernelGGL(spectra_kernel<float>, dim3((unsigned)s->blocks + 1), dim3(FT), 0, (hipStream_t)hip_stream,
                           (const float*)s->raw, s->n, (uint32_t*)mem_dev, (const double*)s->stats, low, zn,
                           (int64_t)(align_up(rows * sizeof(float), 256) / sizeof(float)));
    else
        hipLaunchKernelGGL(spectra_kernel<uint8_t>, dim3((unsigned)s->blocks + 1), dim3(FT), 0, (hipStream_t)hip_stream,
                           (const uint8_t*)s->raw, s->n, (uint32_t*)mem_dev, (const double*)s->stats, low, zn,
                           (int64_t)(align_up(rows * sizeof(float), 256) / sizeof(float)));
    if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
    s->spec = mem_dev;
    s->spec_low = low;
    s->znorm_rest = zn;
    s->norm_stride = (int64_t)(align_up(rows * sizeof(float), 256) / sizeof(float));
    s->spec_bytes = rows * ROW_BYTES;
    return SUSHI_HIP_OK;
}

size_t sushi_hip_batch_bytes(const SushiHipRequest* req_host, int n, int path, int variant, size_t workspace_cap_bytes) try {
    if (!req_host || n <= 0 || (path != SUSHI_HIP_PATH_FFT && path != SUSHI_HIP_PATH_DIRECT)) return 0;
    if (path == SUSHI_HIP_PATH_FFT) variant = direct_variant_count() - 1;
    else if (variant < 0) variant = choose_direct_variant(req_host, n);
    if (variant >= direct_variant_count()) return 0;
    std::vector<SearchDesc> descs;
    int64_t tiles;
    if (make_descs(req_host, n, variant, descs, &tiles) != SUSHI_HIP_OK) return 0;
    if (path == SUSHI_HIP_PATH_DIRECT) return batch_layout(n, path, 0, 0, 0).total;
    const size_t ws = resolve_ws(descs, workspace_cap_bytes);
    Plan plan;
    if (build_plan(descs, ws, plan) != SUSHI_HIP_OK) return 0;
    return batch_layout(n, path, plan.pairs, plan.items.size(), ws).total;
} catch (...) { return 0; }        // std::bad_alloc etc.: nothing crosses the C boundary

int sushi_hip_batch_create(const SushiHipStream* dst, const SushiHipStream* src, const SushiHipRequest* req_host, int n,
                           int path, int variant, size_t workspace_cap_bytes, void* mem_dev, size_t mem_bytes,
                           void* hip_stream, SushiHipBatch** out) try {
    if (!dst || !src || !req_host || !mem_dev || !out || n <= 0) return SUSHI_HIP_EINVAL;
    if (path != SUSHI_HIP_PATH_FFT && path != SUSHI_HIP_PATH_DIRECT) return SUSHI_HIP_EINVAL;
    if (dst->dtype != src->dtype) return SUSHI_HIP_EINVAL;       // cv2.matchTemplate asserts equal types
    if ((uintptr_t)mem_dev & 255) return SUSHI_HIP_EALIGN;
    if (path == SUSHI_HIP_PATH_FFT) {
        if (!dst->spec) return SUSHI_HIP_EINVAL;                 // not searchable: sushi_hip_stream_add_spectra first
        variant = direct_variant_count() - 1;
    } else if (variant < 0) {
        variant = choose_direct_variant(req_host, n);
    }
    if (variant >= direct_variant_count()) return SUSHI_HIP_EINVAL;
    SushiHipBatch* b = new (std::nothrow) SushiHipBatch();
    if (!b) return SUSHI_HIP_EINVAL;
    std::unique_ptr<SushiHipBatch> guard(b);                     // freed on every early return and on an exception
    b->dst = dst; b->src = src; b->n = n; b->path = path; b->variant = variant; b->method = SUSHI_HIP_METHOD_SQDIFF_NORMED;
    b->exclusion = SUSHI_HIP_EXCLUDE_AUTO;
    b->packed_out = nullptr;
    b->host_stats = nullptr; b->stats_ready = nullptr; b->stats_pending = false; b->suspended = 0; b->suspended_at = 0; b->last_suspended = 0;
    b->band = -1; b->last_band = -1; b->band_decided_method = -1; b->band_votes[0] = b->band_votes[1] = 0; b->run_seq = 0; b->audit_every = 2;
    b->bound_model = SUSHI_HIP_BOUND_WORST_CASE;
    {
        // (measurements only, read once per batch: 0 = no excluded pair is audited; "statistical" = round 5's error model)
        const char* e = getenv("SUSHI_HIP_AUDIT_EVERY");
        if (e && *e) { const int v = atoi(e); b->audit_every = v < 0 ? 0 : v; }
        const char* m = getenv("SUSHI_HIP_BOUND_MODEL");
        if (m && !strcmp(m, "statistical")) b->bound_model = SUSHI_HIP_BOUND_STATISTICAL;
    }
    b->mem = (char*)mem_dev; b->last_stream = nullptr; b->ran = false; b->uploaded = nullptr;
    int rc = make_descs(req_host, n, variant, b->descs, &b->n_tiles);
    double flops = 0.0, abytes = 0.0;
    const double width = dst->dtype == SUSHI_HIP_F32 ? 4.0 : 1.0;
    for (int k = 0; k < n && rc == SUSHI_HIP_OK; ++k) {
        const SushiHipRequest& r = req_host[k];
        if (r.tmpl_off + r.tmpl_len > src->n || r.win_start + (int64_t)r.n_pos + r.tmpl_len - 1 > dst->n) rc = SUSHI_HIP_EINVAL;
        flops += 2.0 * (double)r.n_pos * (double)r.tmpl_len;
        abytes += width * ((double)r.n_pos + r.tmpl_len - 1) + width * r.tmpl_len + 8.0;
    }
    b->flops = flops; b->algorithmic_bytes = abytes;
    size_t ws = 0;
    if (rc == SUSHI_HIP_OK && path == SUSHI_HIP_PATH_FFT) {
        ws = resolve_ws(b->descs, workspace_cap_bytes);
        rc = build_plan(b->descs, ws, b->plan);
    }
    if (rc == SUSHI_HIP_OK) {
        b->lay = batch_layout(n, path, b->plan.pairs, b->plan.items.size(), ws);
        if (mem_bytes < b->lay.total) rc = SUSHI_HIP_ENOSPACE;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    if (rc == SUSHI_HIP_OK) {
        // the host arrays live in the handle: the copies may still be in flight when this returns
        if (hipMemcpyAsync(b->mem + b->lay.desc, b->descs.data(), (size_t)n * sizeof(SearchDesc), hipMemcpyHostToDevice, st) != hipSuccess)
            rc = SUSHI_HIP_ELAUNCH;
        if (rc == SUSHI_HIP_OK && !b->plan.order.empty() &&
            hipMemcpyAsync(b->mem + b->lay.order, b->plan.order.data(), b->plan.order.size() * sizeof(int32_t), hipMemcpyHostToDevice, st) != hipSuccess)
            rc = SUSHI_HIP_ELAUNCH;
        if (rc == SUSHI_HIP_OK && !b->plan.items.empty() &&
            hipMemcpyAsync(b->mem + b->lay.items, b->plan.items.data(), b->plan.items.size() * sizeof(int32_t), hipMemcpyHostToDevice, st) != hipSuccess)
            rc = SUSHI_HIP_ELAUNCH;
    }
    // a run may be launched on another stream than this one: it waits for this event first
    if (rc == SUSHI_HIP_OK && (hipEventCreateWithFlags(&b->uploaded, hipEventDisableTiming) != hipSuccess ||
                               hipEventRecord(b->uploaded, st) != hipSuccess))
        rc = SUSHI_HIP_ELAUNCH;
    if (rc != SUSHI_HIP_OK) return rc;
    *out = guard.release();
    return SUSHI_HIP_OK;
} catch (const std::bad_alloc&) { return SUSHI_HIP_ENOMEM; } catch (...) { return SUSHI_HIP_EINTERNAL; }     // nothing crosses the C boundary

int sushi_hip_batch_info(const SushiHipBatch* b, SushiHipBatchInfo* info) {
    if (!b || !info) return SUSHI_HIP_EINVAL;
    memset(info, 0, sizeof(*info));
    info->n_search = b->n; info->path = b->path; info->variant = b->variant;
    info->sub_batches = b->path == SUSHI_HIP_PATH_FFT ? (int32_t)b->plan.subs.size() : 1;
    info->direct_tiles = b->n_tiles;
    info->fft_pairs = b->plan.pairs; info->fft_segments = b->plan.segs;
    info->workspace_bytes = b->plan.ws_bytes; info->mem_bytes = b->lay.total;
    info->flops = b->flops; info->algorithmic_bytes = b->algorithmic_bytes;
    return SUSHI_HIP_OK;
}

void sushi_hip_batch_destroy(SushiHipBatch* b) { delete b; }

int sushi_hip_batch_set_method(SushiHipBatch* b, int method) {
    if (!b) return SUSHI_HIP_EINVAL;
    if (method != SUSHI_HIP_METHOD_SQDIFF_NORMED && method != SUSHI_HIP_METHOD_CCOEFF_NORMED) return SUSHI_HIP_EINVAL;
    b->method = method;                                          // both paths compute both methods
    return SUSHI_HIP_OK;
}

int sushi_hip_batch_set_packed_output(SushiHipBatch* b, int32_t* out_packed_dev) {
    if (!b) return SUSHI_HIP_EINVAL;
    if ((uintptr_t)out_packed_dev & 7) return SUSHI_HIP_EALIGN;
    b->packed_out = out_packed_dev;
    return SUSHI_HIP_OK;
}

int sushi_hip_batch_set_exclusion(SushiHipBatch* b, int mode) {
    if (!b || mode < SUSHI_HIP_EXCLUDE_AUTO || mode > SUSHI_HIP_EXCLUDE_WHOLE) return SUSHI_HIP_EINVAL;
    b->exclusion = mode;
    return SUSHI_HIP_OK;
}

int sushi_hip_batch_set_bound_model(SushiHipBatch* b, int model) {
    if (!b || (model != SUSHI_HIP_BOUND_WORST_CASE && model != SUSHI_HIP_BOUND_STATISTICAL)) return SUSHI_HIP_EINVAL;
    b->bound_model = model;
    return SUSHI_HIP_OK;
}

int sushi_hip_batch_run(SushiHipBatch* b, double delta, int32_t* out_idx_dev, float* out_score_dev, void* hip_stream) try {
    if (!b || !out_idx_dev || !out_score_dev) return SUSHI_HIP_EINVAL;
    hipStream_t st = (hipStream_t)hip_stream;
    const SushiHipStream* dst = b->dst;
    const SushiHipStream* src = b->src;
    const int n_search = b->n;
    StreamRefs r;
    r.dst_xc = dst->xc; r.dst_s1 = dst->s1; r.dst_s2 = dst->s2; r.dst_len = dst->n;
    r.src_xc = src->xc; r.src_s1 = src->s1; r.src_s2 = src->s2; r.src_len = src->n;
    r.centre = sushi_hip_centre(dst->dtype);
    r.dst_raw = dst->raw; r.src_raw = src->raw; r.dtype = dst->dtype;
    const SearchDesc* searches_dev = (const SearchDesc*)(b->mem + b->lay.desc);
    unsigned long long* keys = (unsigned long long*)(b->mem + b->lay.keys);
    b->last_stream = st; b->ran = true; b->direct_pairs = 0;
    if (hipStreamWaitEvent(st, b->uploaded, 0) != hipSuccess) return SUSHI_HIP_ELAUNCH;   // descriptors and plan have landed
    if (b->path == SUSHI_HIP_PATH_DIRECT)
        return launch_direct(r, searches_dev, n_search, (int)b->n_tiles, b->variant, b->method, keys, out_idx_dev,
                             out_score_dev, b->packed_out, st);

    if (!(delta >= 3.8e-6) || delta > 1.0) return SUSHI_HIP_EINVAL;      // the floor covers the scoring arithmetic's own rounding
    unsigned long long* gkeys = keys + n_search;
    int* flags = (int*)(b->mem + b->lay.flags);
    int* flag_list = (int*)(b->mem + b->lay.flag_list);
    int* sub_flagged = (int*)(b->mem + b->lay.sub_flagged);
    RunCounters* counters = (RunCounters*)(b->mem + b->lay.counters);
    const int32_t* order = (const int32_t*)(b->mem + b->lay.order);
    const int32_t* items = (const int32_t*)(b->mem + b->lay.items);
    if (hipMemsetAsync(keys, 0xff, (size_t)2 * n_search * sizeof(uint64_t), st) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    if (hipMemsetAsync(flags, 0, (size_t)n_search * sizeof(int32_t), st) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    if (hipMemsetAsync(counters, 0, sizeof(RunCounters), st) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    ProfCall* pc = nullptr;
    if (g_prof_on) { g_prof.emplace_back(); pc = &g_prof.back(); }

    int* viol = (int*)(b->mem + b->lay.viol);
    if (hipMemsetAsync(viol, 0, (size_t)n_search * sizeof(int32_t), st) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    const unsigned run_seq = b->run_seq++;
    const bool ccm = b->method == SUSHI_HIP_METHOD_CCOEFF_NORMED;
    if (b->stats_pending && hipEventQuery(b->stats_ready) == hipSuccess) {
        b->stats_pending = false;
        const unsigned long long left = b->host_stats[0] - b->host_stats[1];
        if ((double)left > 0.5 * (double)b->plan.pairs) { if (!b->suspended) b->suspended_at = run_seq; b->suspended = 1; }
        else b->suspended = 0;
    }
    // (suspended: every 64th run looks again)
    const bool suspended_now = b->exclusion == SUSHI_HIP_EXCLUDE_AUTO && b->suspended && ((run_seq - b->suspended_at) & 63u) != 63u;
    b->last_suspended = suspended_now ? 1 : 0;
    bool excluded_any = false;

    // (A two-stream variant that overlapped the multiply-accumulate of sub-batch n+1 with the inverse
    // transforms of sub-batch n was measured 5 % slower: both kernels only contend.)
    for (const SubBatch& sbt : b->plan.subs) {
        const int n_sub = sbt.b0 - sbt.a0;
        const WsLayout wl = ws_layout(sbt.pairs, sbt.segs, n_sub);
        char* wsp = b->mem + b->lay.ws;
        uint32_t* tspec = (uint32_t*)(wsp + wl.tspec);
        uint4* y = (uint4*)(wsp + wl.y);
        unsigned long long* cand = (unsigned long long*)(wsp + wl.cand);
        int* pairmap = (int*)(wsp + wl.pairmap);
        float* pair_lb = (float*)(wsp + wl.pair_lb);
        TemplConsts* tconst = (TemplConsts*)(wsp + wl.tconst);
        TileDesc* tiles = (TileDesc*)(wsp + wl.tiles);
        int32_t* candbuf = (int32_t*)(wsp + wl.candbuf);
        uint4* tspec_low = (uint4*)(wsp + wl.tspec_low);
        uint4* ylow = (uint4*)(wsp + wl.ylow);
        float* tnorm_rest = (float*)(wsp + wl.tnorm_rest);
        int* scount = (int*)(wsp + wl.scount);

        hipEvent_t t0 = prof_begin(pc, st);
        TspecArgs ta;
        ta.src_raw = src->raw; ta.searches = searches_dev + sbt.a0; ta.n_sub = n_sub; ta.sub_first_seg = sbt.first_seg;
        ta.tspec = tspec; ta.sub_first_pair = sbt.first_pair; ta.pairmap = pairmap; ta.tconst = tconst;
        ta.src_s1 = src->s1; ta.src_s2 = src->s2; ta.centre = r.centre; ta.dst_stats = dst->stats; ta.method = b->method;
        ta.tspec_low = tspec_low; ta.tnorm_rest = tnorm_rest;
        if (hipMemsetAsync(tnorm_rest, 0, (size_t)sbt.segs * sizeof(float), st) != hipSuccess) return SUSHI_HIP_ELAUNCH;
        if (src->dtype == SUSHI_HIP_F32) hipLaunchKernelGGL(tspec_kernel<float>, dim3((unsigned)sbt.segs), dim3(FT), 0, st, ta);
        else hipLaunchKernelGGL(tspec_kernel<uint8_t>, dim3((unsigned)sbt.segs), dim3(FT), 0, st, ta);
        if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
        prof_end(pc, t0, SUSHI_HIP_STAGE_TSPEC, st);

        // The exclusion costs a pass over Y (~16 ns per pair) and half a dozen launches (~60 us); transforming a pair ~37 ns:
        // it pays from ~3000 pairs on, plus two per search (the pairs transformed first are transformed either way).
        const bool exclude = b->exclusion == SUSHI_HIP_EXCLUDE_ALWAYS || b->exclusion == SUSHI_HIP_EXCLUDE_BAND ||
                             b->exclusion == SUSHI_HIP_EXCLUDE_WHOLE ||
                             (b->exclusion == SUSHI_HIP_EXCLUDE_AUTO && !suspended_now && sbt.pairs > 3000 + 2 * (int64_t)n_sub);
        excluded_any = excluded_any || exclude;
        BoundArgs ba;
        memset(&ba, 0, sizeof(ba));
        ba.dst_stats = dst->stats; ba.searches = searches_dev + sbt.a0; ba.sub_first_pair = sbt.first_pair;
        ba.first_search = sbt.a0; ba.dst_len = dst->n; ba.pairmap = pairmap; ba.tconst = tconst; ba.ubase = dst->base;
        ba.sbase = dst->base + (dst->blocks + 1); ba.nb = dst->blocks; ba.coarse = dst->coarse; ba.nc = dst->nc;
        ba.slb = (float*)(wsp + wl.slb); ba.n_sub = n_sub; ba.n_pairs = (int)sbt.pairs; ba.plist = (int*)(wsp + wl.plist);
        ba.slist = (int*)(wsp + wl.slist); ba.scount = scount; ba.order = order + sbt.first_pair;
        ba.gkeys = gkeys; ba.pair_lb = pair_lb; ba.counters = counters;
        ba.acc = (float*)(wsp + wl.acc);
        ba.sub_first_seg = sbt.first_seg; ba.tnorm_rest = tnorm_rest; ba.znorm_rest = dst->znorm_rest; ba.norm_stride = dst->norm_stride; ba.band_votes = scount + 2;
        ba.audit_mark = (unsigned char*)(wsp + wl.audit_mark); ba.audit_seq = run_seq; ba.audit_every = b->audit_every;
        // What a packed-half transform output (bound_low_kernel / bound_kernel) may be off by, in units of the largest pass-1 value:
        // every output is a sum of 64 pass-1 values through ROUNDING LEVELS of 2^-11 each -- a level at which the partial sums hold m
        // terms each costs 2^-11 m per value and 64 / m values meet in an output: 2^-11 64 per level whatever m.  Worst path: pass 1's
        // own result 1, its half-precision matrix (2^-12 sqrt 2 per entry, sum |inputs| <= 4 max |output| by Parseval) 2.8, pass 2's
        // twiddle 2 + its radix-16 butterflies 1 + 1 + 2 + 3 (h_bfly_root32: q = 0 / 8 one rounding, 4 / 12 two, others three on the
        // e - w o side), pass 3's twiddle 2 + radix 4: 1 + 1, four levels of half-precision twiddle constants at 2^-12 each = 2:
        // 18.8 levels = 0.59.  (Round 5's 0.29 counted 9: about right for independent roundings, not a worst case.)
        ba.worst_case = b->bound_model == SUSHI_HIP_BOUND_WORST_CASE ? 1 : 0;
        ba.half_err = ba.worst_case ? 0.6f : 0.29f;
        auto launch_slb = [&](const BoundArgs& x) {
            if (ccm) hipLaunchKernelGGL(slb_kernel<SUSHI_HIP_METHOD_CCOEFF_NORMED>, dim3((unsigned)((sbt.pairs + 3) / 4)), dim3(256), 0, st, x);
            else hipLaunchKernelGGL(slb_kernel<SUSHI_HIP_METHOD_SQDIFF_NORMED>, dim3((unsigned)((sbt.pairs + 3) / 4)), dim3(256), 0, st, x);
            return launch_ok();
        };

        // Which form of the exclusion: the band-split form multiplies, stores and transforms only the low band of
        // every spectrum and bounds the rest by the rows' norms -- a quarter of the bytes and a third of the instructions, IF the
        // streams keep most of their energy in the band (audio does; white noise does not).  Decided once per batch and method, on
        // the device's own numbers: with nothing at all from the low band, does the rest alone leave the bound room to exclude?
        // (One small kernel over the first sub-batch's pairs and one 8-byte read-back, in the first run only.)
        int band = 0;
        b->last_band = -1;
        if (exclude) {
            if (b->exclusion == SUSHI_HIP_EXCLUDE_BAND) band = 1;
            else if (b->exclusion == SUSHI_HIP_EXCLUDE_WHOLE) band = 0;
            else {
                if (b->band < 0 || b->band_decided_method != b->method) {
                    if (hipMemsetAsync(scount + 2, 0, 2 * sizeof(int), st) != hipSuccess) return SUSHI_HIP_ELAUNCH;
                    BoundArgs bp = ba;
                    bp.band = 2;
                    if (launch_slb(bp) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
                    if (hipMemcpyAsync(b->band_votes, scount + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
                        hipStreamSynchronize(st) != hipSuccess)
                        return SUSHI_HIP_ELAUNCH;
                    // (measured at BASELINE configs[2]: 97 % of the pairs vote for it at 12 dB of noise on the source -- 9.7 ms against 17.5 for
                    // the whole-row form --, 87 % at 6 dB -- 12.5 against 17.5 --, 14 % at 0 dB -- 28.7 against 18.7)
                    b->band = b->band_votes[0] > 0 && (double)b->band_votes[1] >= 0.75 * (double)b->band_votes[0] ? 1 : 0;
                    b->band_decided_method = b->method;
                }
                band = b->band;
            }
            b->last_band = band;
        }

        // the multiply-accumulate over ALL pairs: of the low rows (band-split form) or of whole rows; `enable`: a device flag that
        // may call the launch off (the whole-row launch queued behind the survivors' list, below)
        auto launch_mac = [&](const bool low, const int* enable) {
            MacArgs ma;
            ma.spec_blocks = dst->blocks;
            ma.searches = searches_dev + sbt.a0; ma.tconst = tconst; ma.sub_first_seg = sbt.first_seg;
            ma.sub_first_pair = sbt.first_pair;
            ma.dummy = (uint4*)(wsp + wl.dummy);
            ma.enable = enable;
            if (low) { ma.spec = (const uint4*)dst->spec_low; ma.tspec = tspec_low; ma.y = ylow; }
            else { ma.spec = (const uint4*)dst->spec; ma.tspec = (const uint4*)tspec; ma.y = y; }
            for (int kern = 0; kern < 2; ++kern) {
                if (sbt.item_count[kern] == 0) continue;
                ma.items = items + (size_t)sbt.item_first[kern] * (1 + MAC_SPW);
                ma.n_items = sbt.item_count[kern];
                const int chunks = (low ? LROWE : ROWE) / MAC_BW;
                ma.chunk_group = std::min(sbt.chunk_group[kern], chunks / 8);
                const dim3 grid((unsigned)chunks * (unsigned)ma.n_items);
                if (low) {
                    if (kern == 0) hipLaunchKernelGGL(mac_kernel<LROWE>, grid, dim3(MAC_THREADS), 0, st, ma);
                    else hipLaunchKernelGGL(mac_long_kernel<LROWE>, grid, dim3(MAC_THREADS), 0, st, ma);
                } else {
                    if (kern == 0) hipLaunchKernelGGL(mac_kernel<ROWE>, grid, dim3(MAC_THREADS), 0, st, ma);
                    else hipLaunchKernelGGL(mac_long_kernel<ROWE>, grid, dim3(MAC_THREADS), 0, st, ma);
                }
                if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            }
            return SUSHI_HIP_OK;
        };
        t0 = prof_begin(pc, st);
        if (launch_mac(band != 0, nullptr) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
        prof_end(pc, t0, SUSHI_HIP_STAGE_MAC, st);

        t0 = prof_begin(pc, st);
        // the candidate rows start empty: ifft_kernel writes only the entries that exist
        if (hipMemsetAsync(cand, 0xff, (size_t)sbt.pairs * FFT_ROW * sizeof(unsigned long long), st) != hipSuccess)
            return SUSHI_HIP_ELAUNCH;
        IfftArgs ia;
        memset(&ia, 0, sizeof(ia));
        ia.y = (const uint2*)y; ia.dst_stats = dst->stats; ia.searches = searches_dev + sbt.a0; ia.n_sub = n_sub; ia.first_search = sbt.a0;
        ia.sub_first_pair = sbt.first_pair; ia.dst_len = dst->n; ia.delta = (float)delta; ia.cand = cand; ia.pair_lb = pair_lb; ia.gkeys = gkeys;
        ia.pairmap = pairmap; ia.tconst = tconst; ia.order = order + sbt.first_pair;
        ia.urel = dst->urel; ia.nb = dst->blocks; ia.ubase = dst->base;
        ia.usrel = dst->usrel; ia.sbase = dst->base + (dst->blocks + 1);
        ia.flags = flags; ia.flag_list = flag_list; ia.sub_flagged = sub_flagged; ia.tiles = tiles; ia.candbuf = candbuf;
        ia.cand_cap = (int)cand_capacity(sbt.pairs); ia.counters = counters;
        ia.viol = viol;
        auto launch_ifft = [&](const IfftArgs& x, unsigned grid) {
            if (x.count && !x.list_direct) {                         // a list whose length only the device knows: a fixed grid strides over its tail
                if (ccm) hipLaunchKernelGGL(ifft_list_kernel<SUSHI_HIP_METHOD_CCOEFF_NORMED>, dim3(grid), dim3(FT), 0, st, x);
                else hipLaunchKernelGGL(ifft_list_kernel<SUSHI_HIP_METHOD_SQDIFF_NORMED>, dim3(grid), dim3(FT), 0, st, x);
            } else {
                if (ccm) hipLaunchKernelGGL(ifft_kernel<SUSHI_HIP_METHOD_CCOEFF_NORMED>, dim3(grid), dim3(FT), 0, st, x);
                else hipLaunchKernelGGL(ifft_kernel<SUSHI_HIP_METHOD_SQDIFF_NORMED>, dim3(grid), dim3(FT), 0, st, x);
            }
            return launch_ok();
        };
        // the whole rows of LISTED pairs (band-split form: nothing but the low band exists until a pair is to be transformed)
        auto launch_mac_list = [&](const int* list, const int* count, int n_list, const int* disable, int long_only) {
            MacListArgs la;
            la.disable = disable; la.long_only = long_only;
            la.spec = (const uint4*)dst->spec; la.spec_blocks = dst->blocks; la.tspec = (const uint4*)tspec; la.y = y;
            la.searches = searches_dev + sbt.a0; la.tconst = tconst; la.pairmap = pairmap; la.list = list; la.count = count;
            la.n_list = n_list; la.sub_first_seg = sbt.first_seg; la.sub_first_pair = sbt.first_pair;
            const int64_t want = (int64_t)n_list * MACL_PARTS;
            hipLaunchKernelGGL(mac_list_kernel, dim3((unsigned)std::min<int64_t>(want, 256 * 32)), dim3(MACL_THREADS), 0, st, la);
            return launch_ok();
        };
        if (!exclude) {
            // every pair, in the L2-friendly schedule (what round 3 did for every batch)
            if (launch_ifft(ia, (unsigned)sbt.pairs) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            b->direct_pairs += sbt.pairs;
        } else {
            // A lower bound of every pair's scores first (three of the transform's four passes, no scoring); then the most
            // promising pair of every search, which leaves the search's threshold; then whatever the bound could not exclude
            // (header of bound_kernel)
            ba.band = band;
            ba.y = band ? (const uint2*)ylow : (const uint2*)y;
            if (hipMemsetAsync(ba.acc, 0, (size_t)sbt.pairs * 2 * sizeof(float), st) != hipSuccess) return SUSHI_HIP_ELAUNCH;
            {
                // persistent waves: four workgroups of four per CU (the kernel's register budget), fewer for a small batch
                const int per_pair = band ? 1 : 16;                  // bound_low_kernel: a wave per pair
                const int64_t want = (sbt.pairs * per_pair + BOUND_THREADS / 64 - 1) / (BOUND_THREADS / 64);
                const unsigned grid = (unsigned)std::min<int64_t>(want, 256 * (band ? 3 : 4));   // (what is resident at each kernel's registers)
                if (band) hipLaunchKernelGGL(bound_low_kernel, dim3(grid), dim3(BOUND_THREADS), 0, st, ba);
                else hipLaunchKernelGGL(bound_kernel, dim3(grid), dim3(BOUND_THREADS), 0, st, ba);
                if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
                if (launch_slb(ba) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            }
            prof_end(pc, t0, SUSHI_HIP_STAGE_BOUND, st);
            t0 = prof_begin(pc, st);
            hipLaunchKernelGGL(pilot_kernel, dim3((unsigned)n_sub), dim3(64), 0, st, ba);
            if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            IfftArgs ip = ia;
            ip.slb = ba.slb; ip.audit_mark = nullptr;              // (the pairs transformed first are nobody's excluded pairs)
            ip.order = ba.plist; ip.count = nullptr;
            if (band && launch_mac_list(ba.plist, nullptr, n_sub, nullptr, 0) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            if (launch_ifft(ip, (unsigned)n_sub) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            hipLaunchKernelGGL(survivor_kernel, dim3((unsigned)((sbt.pairs + 255) / 256)), dim3(256), 0, st, ba);
            if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            const int* final_list = ba.slist;
            const int* final_count = ba.scount;
            if (band) {
                // the second look at what the bound left (header of bound_low_exact_kernel): sharper bound, shorter list
                ba.list = ba.slist; ba.list_count = ba.scount;
                ba.list2 = (int*)(wsp + wl.slist2); ba.list2_count = scount + 5;
                if (hipMemsetAsync(scount + 5, 0, sizeof(int), st) != hipSuccess) return SUSHI_HIP_ELAUNCH;
                hipLaunchKernelGGL(bound_low_exact_kernel, dim3(256 * 4), dim3(BLE_T), 0, st, ba);
                if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
                if (ccm) hipLaunchKernelGGL(slb_list_kernel<SUSHI_HIP_METHOD_CCOEFF_NORMED>, dim3(256 * 2), dim3(256), 0, st, ba);
                else hipLaunchKernelGGL(slb_list_kernel<SUSHI_HIP_METHOD_SQDIFF_NORMED>, dim3(256 * 2), dim3(256), 0, st, ba);
                if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
                hipLaunchKernelGGL(survivor2_kernel, dim3(256), dim3(256), 0, st, ba);
                if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
                final_list = ba.list2; final_count = ba.list2_count;
                // whole rows of what is left: pair by pair while few are (the usual case), by the dense multiply-accumulate over all
                // pairs when the bound excluded little (searches without a match: nothing can be excluded) -- decided on the device
                int* dense = scount + 4;
                hipLaunchKernelGGL(dense_mode_kernel, dim3(1), dim3(1), 0, st, final_count, (int)sbt.pairs, dense);
                if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
                {
                    MacRowsArgs ra;
                    ra.spec = (const uint4*)dst->spec; ra.spec_blocks = dst->blocks; ra.tspec = (const uint4*)tspec; ra.y = y;
                    ra.searches = searches_dev + sbt.a0; ra.tconst = tconst; ra.mark = ba.audit_mark; ra.n_sub = n_sub;
                    ra.sub_first_seg = sbt.first_seg; ra.sub_first_pair = sbt.first_pair; ra.disable = dense;
                    const int64_t want = (int64_t)n_sub * MACL_PARTS;
                    if (sbt.item_count[0] > 0) hipLaunchKernelGGL(mac_rows_kernel<0>, dim3((unsigned)std::min<int64_t>(want, 256 * 32)), dim3(MACL_THREADS), 0, st, ra);
                    if (sbt.item_count[1] > 0) hipLaunchKernelGGL(mac_rows_kernel<1>, dim3((unsigned)std::min<int64_t>(want, 256 * 16)), dim3(MACL_THREADS), 0, st, ra);
                    if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
                }
                if (sbt.long_patterns && launch_mac_list(final_list, final_count, (int)sbt.pairs, dense, 1) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
                if (launch_mac(false, dense) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            }
            ip.order = final_list; ip.count = final_count; ip.audit_mark = ba.audit_mark;
            // One workgroup per list slot up to what the list usually holds (an eighth of the pairs: empty slots there cost a
            // workgroup's launch each, ~1 ns), and a fixed grid striding over whatever lies beyond: the striding form alone runs
            // at half the rate per pair (the loop costs it registers), one workgroup per POSSIBLE slot cost 0.3 ms of empty launches.
            const unsigned direct = (unsigned)std::min<int64_t>(sbt.pairs, std::max<int64_t>(4096, sbt.pairs / 8));
            ip.list_first = 0; ip.list_direct = 1;
            if (launch_ifft(ip, direct) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            if ((int64_t)direct < sbt.pairs) {
                ip.list_first = (int)direct; ip.list_direct = 0;
                if (launch_ifft(ip, (unsigned)std::min<int64_t>(sbt.pairs - direct, 1024)) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            }
        }
        prof_end(pc, t0, SUSHI_HIP_STAGE_IFFT, st);

        t0 = prof_begin(pc, st);
        RefineParams rp;
        rp.r = r; rp.searches = searches_dev; rp.first_search = sbt.a0; rp.n_sub = n_sub; rp.sub_first_pair = sbt.first_pair;
        rp.cand = cand; rp.pair_lb = pair_lb; rp.gkeys = gkeys; rp.keys = keys; rp.flags = flags; rp.flag_list = flag_list;
        rp.sub_flagged = sub_flagged; rp.counters = counters; rp.delta = (float)delta; rp.method = b->method;
        rp.citems = (int*)(wsp + wl.citems); rp.n_citems = (int*)(wsp + wl.scount) + 1;
        rp.viol = viol;
        ia.citems = rp.citems; ia.n_citems = rp.n_citems;
        int rc = launch_refine(rp, st);
        if (rc != SUSHI_HIP_OK) return rc;
        prof_end(pc, t0, SUSHI_HIP_STAGE_REFINE, st);

        // searches the lists could not finish: collect their candidates per tile, evaluate those exactly
        // (both kernels leave after one load when nothing is flagged)
        t0 = prof_begin(pc, st);
        if (b->method == SUSHI_HIP_METHOD_CCOEFF_NORMED)
            hipLaunchKernelGGL(collect_kernel<SUSHI_HIP_METHOD_CCOEFF_NORMED>, dim3(COLLECT_GRID), dim3(FT), 0, st, ia);
        else
            hipLaunchKernelGGL(collect_kernel<SUSHI_HIP_METHOD_SQDIFF_NORMED>, dim3(COLLECT_GRID), dim3(FT), 0, st, ia);
        if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
        TileParams tp;
        tp.r = r; tp.searches = searches_dev; tp.tiles = tiles; tp.cand = candbuf; tp.keys = keys; tp.counters = counters;
        tp.method = b->method;
        rc = launch_tiles(tp, st);
        if (rc != SUSHI_HIP_OK) return rc;
        prof_end(pc, t0, SUSHI_HIP_STAGE_FINISH, st);
    }
    hipEvent_t t0 = prof_begin(pc, st);
    const int rc = launch_unpack(keys, n_search, b->method, out_idx_dev, out_score_dev, b->packed_out, st);
    prof_end(pc, t0, SUSHI_HIP_STAGE_FINISH, st);
    if (rc == SUSHI_HIP_OK && excluded_any && b->exclusion == SUSHI_HIP_EXCLUDE_AUTO && !b->stats_pending) {
        // what this run's exclusion left, for the runs after it (never waited for: the event is queried)
        if (!b->host_stats && hipHostMalloc((void**)&b->host_stats, 2 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) b->host_stats = nullptr;
        if (b->host_stats && !b->stats_ready && hipEventCreateWithFlags(&b->stats_ready, hipEventDisableTiming) != hipSuccess) b->stats_ready = nullptr;
        if (b->host_stats && b->stats_ready &&
            hipMemcpyAsync(b->host_stats, &counters->pairs_transformed, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st) == hipSuccess &&
            hipEventRecord(b->stats_ready, st) == hipSuccess)
            b->stats_pending = true;
    }
    return rc;
} catch (const std::bad_alloc&) { return SUSHI_HIP_ENOMEM; } catch (...) { return SUSHI_HIP_EINTERNAL; }     // nothing crosses the C boundary

int sushi_hip_batch_diagnostics(SushiHipBatch* b, SushiHipBatchDiag* diag, float* ranking_err_host, int32_t* flagged_host) try {
    if (!b || !diag) return SUSHI_HIP_EINVAL;
    memset(diag, 0, sizeof(*diag));
    if (!b->ran || b->path != SUSHI_HIP_PATH_FFT) {
        if (ranking_err_host) memset(ranking_err_host, 0, (size_t)b->n * sizeof(float));
        if (flagged_host) memset(flagged_host, 0, (size_t)b->n * sizeof(int32_t));
        return SUSHI_HIP_OK;
    }
    if (hipStreamSynchronize(b->last_stream) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    RunCounters c;
    if (hipMemcpy(&c, b->mem + b->lay.counters, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    diag->flagged = c.n_flagged; diag->all_positions = c.n_all_positions;
    diag->tiles_dense = (int64_t)c.tiles_dense; diag->tiles_sparse = (int64_t)c.tiles_sparse;
    diag->candidates = (int64_t)c.candidates;
    memcpy(&diag->max_bound_ratio, &c.max_ratio_bits, sizeof(float));
    memcpy(&diag->max_bound_ratio_noncandidate, &c.max_ratio_audit_bits, sizeof(float));
    diag->audited = (int64_t)c.audited;
    diag->pairs_transformed = (int64_t)c.pairs_transformed + b->direct_pairs;
    diag->excluded_audited = (int64_t)c.excluded_audited;
    memcpy(&diag->max_slb_ratio_excluded, &c.max_slb_ratio_bits, sizeof(float));
    diag->slb_violations = c.slb_violations;
    diag->band = b->last_band;
    diag->suspended = b->last_suspended;
    diag->band_votes[0] = b->band_votes[0]; diag->band_votes[1] = b->band_votes[1];
    diag->second_look_audited = (int64_t)c.second_look_audited;
    std::vector<int32_t> fl((size_t)b->n);
    if (hipMemcpy(fl.data(), b->mem + b->lay.flags, (size_t)b->n * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess)
        return SUSHI_HIP_ELAUNCH;
    if (flagged_host) memcpy(flagged_host, fl.data(), (size_t)b->n * sizeof(int32_t));
    if (ranking_err_host) {
        std::vector<unsigned long long> g((size_t)b->n);
        if (hipMemcpy(g.data(), b->mem + b->lay.keys + (size_t)b->n * sizeof(unsigned long long),
                      (size_t)b->n * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
            return SUSHI_HIP_ELAUNCH;
        for (int k = 0; k < b->n; ++k) {
            const uint32_t bits = fl[k] ? 0u : (uint32_t)(g[k] & 0xffffffffull);     // flagged searches keep their threshold there
            memcpy(&ranking_err_host[k], &bits, sizeof(float));
        }
    }
    return SUSHI_HIP_OK;
} catch (const std::bad_alloc&) { return SUSHI_HIP_ENOMEM; } catch (...) { return SUSHI_HIP_EINTERNAL; }     // nothing crosses the C boundary

int sushi_hip_batch_pair_bounds(SushiHipBatch* b, float* slb_host, float* acc_host, int64_t* n_pairs) try {
    if (!b || !n_pairs) return SUSHI_HIP_EINVAL;
    if (!b->ran || b->path != SUSHI_HIP_PATH_FFT || b->plan.subs.empty()) { *n_pairs = 0; return SUSHI_HIP_OK; }
    if (hipStreamSynchronize(b->last_stream) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    const SubBatch& sbt = b->plan.subs.back();
    const WsLayout wl = ws_layout(sbt.pairs, sbt.segs, sbt.b0 - sbt.a0);
    const int64_t cap = *n_pairs;
    *n_pairs = sbt.pairs;
    if (cap < sbt.pairs) return SUSHI_HIP_ENOSPACE;
    const char* wsp = b->mem + b->lay.ws;
    if (slb_host && hipMemcpy(slb_host, wsp + wl.slb, (size_t)sbt.pairs * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    if (acc_host && hipMemcpy(acc_host, wsp + wl.acc, (size_t)sbt.pairs * 2 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    return SUSHI_HIP_OK;
} catch (const std::bad_alloc&) { return SUSHI_HIP_ENOMEM; } catch (...) { return SUSHI_HIP_EINTERNAL; }

int sushi_hip_batch_workspace_view(SushiHipBatch* b, int which, void** ptr_dev, size_t* bytes) {
    if (!b || !ptr_dev || !bytes) return SUSHI_HIP_EINVAL;
    *ptr_dev = nullptr; *bytes = 0;
    if (!b->ran || b->path != SUSHI_HIP_PATH_FFT || b->plan.subs.empty()) return SUSHI_HIP_OK;
    if (hipStreamSynchronize(b->last_stream) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    const SubBatch& sbt = b->plan.subs.back();
    const WsLayout wl = ws_layout(sbt.pairs, sbt.segs, sbt.b0 - sbt.a0);
    char* wsp = b->mem + b->lay.ws;
    switch (which) {
        case SUSHI_HIP_WS_TSPEC: *ptr_dev = wsp + wl.tspec; *bytes = (size_t)sbt.segs * ROW_BYTES; break;
        case SUSHI_HIP_WS_Y: *ptr_dev = wsp + wl.y; *bytes = (size_t)sbt.pairs * ROW_BYTES; break;
        case SUSHI_HIP_WS_TSPEC_LOW: *ptr_dev = wsp + wl.tspec_low; *bytes = (size_t)sbt.segs * LROW_BYTES; break;
        case SUSHI_HIP_WS_Y_LOW: *ptr_dev = wsp + wl.ylow; *bytes = (size_t)sbt.pairs * LROW_BYTES; break;
        default: return SUSHI_HIP_EINVAL;
    }
    return SUSHI_HIP_OK;
}

int sushi_hip_profile_begin(void) {
    for (ProfCall& c : g_prof)
        for (ProfSpan& sp : c.spans) { (void)hipEventDestroy(sp.t0); (void)hipEventDestroy(sp.t1); }
    g_prof.clear();
    g_prof_on = true;
    return SUSHI_HIP_OK;
}

int sushi_hip_profile_end(float* stage_ms, int max_calls, int* n_calls) try {
    g_prof_on = false;
    if (!stage_ms || !n_calls || max_calls < 0) return SUSHI_HIP_EINVAL;
    int out = 0;
    int rc = SUSHI_HIP_OK;
    for (ProfCall& c : g_prof) {
        if (out < max_calls && !c.spans.empty()) {
            float* row = stage_ms + (size_t)out * SUSHI_HIP_NSTAGES;
            for (int k = 0; k < SUSHI_HIP_NSTAGES; ++k) row[k] = 0.f;
            for (ProfSpan& sp : c.spans) {
                float ms = 0.f;
                if (hipEventSynchronize(sp.t1) != hipSuccess || hipEventElapsedTime(&ms, sp.t0, sp.t1) != hipSuccess) {
                    rc = SUSHI_HIP_ELAUNCH;
                    break;
                }
                if (sp.stage >= 0 && sp.stage < SUSHI_HIP_NSTAGES) row[sp.stage] += ms;
            }
            ++out;
        }
        for (ProfSpan& sp : c.spans) { (void)hipEventDestroy(sp.t0); (void)hipEventDestroy(sp.t1); }
    }
    g_prof.clear();
    *n_calls = out;
    return rc;
} catch (const std::bad_alloc&) { return SUSHI_HIP_ENOMEM; } catch (...) { return SUSHI_HIP_EINTERNAL; }     // nothing crosses the C boundary

}  // extern "C"
