// sushi_amd/csrc/sushi_common.hpp -- device helpers shared by the direct (MFMA) and FFT paths.
#ifndef SUSHI_COMMON_HPP
#define SUSHI_COMMON_HPP

#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "../../include/sushi_hip.h"

namespace sushi {

// ---- overlap-save geometry ------------------------------------------------------------
// Transform length N = 2^FFT_LOGN complex points; patterns are cut into segments of FFT_SEG samples, so a real
// block of N samples yields FFT_H = N - FFT_SEG valid positions; two real blocks FFT_H apart are packed into one
// complex block (a "pair": 2 * FFT_H positions per transform).  Spectra are kept at every multiple of FFT_SEG
// ("block" j = samples from j * FFT_SEG on); consecutive pairs of a search are FFT_STEP blocks apart.
constexpr int FFT_LOGN = 14;
constexpr int FFT_N = 1 << FFT_LOGN;
constexpr int FFT_SEG = 4096;
constexpr int FFT_HOP = FFT_SEG;                   // also the block size of the relative window-energy prefix (urel / base)
constexpr int FFT_VB = FFT_N / FFT_SEG - 1;        // valid blocks per half of a pair: 3
constexpr int FFT_H = FFT_VB * FFT_SEG;            // result positions per half
constexpr int FFT_STEP = 2 * FFT_VB;               // blocks between consecutive pairs
constexpr int FFT_CAND = 8;                        // candidate slots per block pair (+ overflow marker + error bound + audit positions)
constexpr int FFT_AUDIT = 4;                       // positions of an audit run: consecutive (one exact evaluation's worth of loads)
constexpr int AUDIT_RUNS = 4;                      // audit runs a transformed pair leaves, and audit runs refine_kernel evaluates per search
constexpr int FFT_ROW = 32;                        // 64-bit entries per pair in the candidate array: two 128-byte lines
static_assert(FFT_CAND + 2 + AUDIT_RUNS * FFT_AUDIT <= FFT_ROW, "candidates, overflow marker, error bound, audit runs");
constexpr int TILE = 1024;                         // positions per exact-evaluation tile (aligned to the absolute grid)
constexpr int TILES_PER_PAIR = 2 * FFT_H / TILE;
// error model of the f32 FFT stage: |corr_f32 - corr| <= FFT_KE * 2^-24 * |T| * |Z|, Z = the samples that enter the
// pair's transforms (n_seg + 2 * FFT_VB blocks).  Calibrated: the largest ratio measured over the parity and property
// tests is recorded by refine_kernel (diagnostics) and stays below FFT_KE / 3; a candidate whose exact score
// violates its bound sends the whole search to exact evaluation.
constexpr int COARSE_G = 256;                      // granularity of the coarse prefix table (SushiHipStream.coarse)
constexpr float FFT_KE = 32.0f;
constexpr unsigned long long NO_KEY = ~0ull;

// XCD-aware remap (MI355X: block b runs on XCD b % 8): give every XCD a contiguous run of
// logical tiles so that the tiles of one search (same template, overlapping windows) share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, k = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

__device__ __forceinline__ unsigned long long shfl_down_u64(unsigned long long v, int d) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_down(lo, d, 64);
    hi = __shfl_down(hi, d, 64);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = shfl_down_u64(v, d);
        v = o < v ? o : v;
    }
    return v;                                   // valid in lane 0
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    return v;
}

// exclusive scan over one wave (64 lanes) of doubles; returns exclusive prefix, *total = wave sum
__device__ __forceinline__ double wave_excl_scan(double v, double* total) {
    const int lane = threadIdx.x & 63;
    double incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    *total = __shfl(incl, 63, 64);
    return incl - v;
}

__device__ __forceinline__ unsigned long long make_key(float score, unsigned pos) {
    return ((unsigned long long)__float_as_uint(score) << 32) | pos;      // scores are >= 0: uint order == float order
}
__device__ __forceinline__ float key_score(unsigned long long key) { return __uint_as_float((unsigned)(key >> 32)); }
__device__ __forceinline__ unsigned key_pos(unsigned long long key) { return (unsigned)(key & 0xffffffffull); }
// arg-MAX through the same atomicMin (TM_CCOEFF_NORMED, scores in [-1, 1]): the key holds -score under the usual
// order-preserving map of float bits to unsigned, so the smallest key is the largest score at its lowest position.
__device__ __forceinline__ unsigned long long make_key_max(float score, unsigned pos) {
    const unsigned b = __float_as_uint(-(score + 0.0f));                  // + 0.0f: -0.0 and 0.0 tie, as in NumPy
    const unsigned u = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)u << 32) | pos;
}
__device__ __forceinline__ float key_score_max(unsigned long long key) {
    const unsigned u = (unsigned)(key >> 32);
    const unsigned b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return -__uint_as_float(b);
}

// Template statistics in the order cv2 derives them (templmatch.cpp common_matchTemplate:
// meanStdDev -> templSum2 / templNorm), from the float64 prefix sums of the source stream's samples
// (s1 = sum x, s2 = sum x^2 of the samples as they are: exact for uint8, correctly rounded for float32).
struct TemplStats {
    double tS1;           // sum T
    double cM;            // centre^2 * M (what the direct kernel's centred cross term has to be corrected by)
    double tU, tnorm;     // templSum2, templNorm as cv2 has them (TM_SQDIFF_NORMED: numType != 1)
    double tmean, tnorm_c; // templMean, templNorm of TM_CCOEFF_NORMED (numType == 1)
    bool flat;            // templSdv^2 < DBL_EPSILON: cv2 returns a result of all ones for TM_CCOEFF_NORMED
};

__device__ __forceinline__ TemplStats templ_stats(const double* __restrict__ s1, const double* __restrict__ s2,
                                                  int64_t off, int M, double centre) {
    TemplStats t;
    t.cM = centre * centre * (double)M;
    t.tS1 = s1[off + M] - s1[off];
    const double t_sq = s2[off + M] - s2[off];                       // sum T^2
    const double invArea = 1.0 / (double)M;
    const double t_mean = t.tS1 * invArea;
    double t_var = t_sq * invArea - t_mean * t_mean;
    t_var = t_var > 0.0 ? t_var : 0.0;
    const double t_sdv = sqrt(t_var);
    const double t_norm2 = t_sdv * t_sdv + t_mean * t_mean;          // templSum2 before "/= invArea"
    t.tU = t_norm2 / invArea;                                        // templSum2
    t.tnorm = sqrt(t_norm2) / sqrt(invArea);                         // templNorm
    t.tmean = t_mean;
    t.flat = t_sdv * t_sdv < DBL_EPSILON;
    t.tnorm_c = sqrt(t_sdv * t_sdv) / sqrt(invArea);
    return t;
}

// OpenCV templmatch.cpp common_matchTemplate(), TM_SQDIFF_NORMED branch, one position.
// corr_u: sum T*I, wU: sum I^2 over the window, tU: sum T^2, tnorm: sqrt(tU).
__device__ __forceinline__ float finish_sqdiff_normed(double corr_u, double wU, double tU, double tnorm) {
    double num = (double)(float)corr_u;          // cv2 keeps corr in its float32 result Mat
    num = wU - 2.0 * num + tU;
    num = num > 0.0 ? num : 0.0;
    const double diff2 = wU > 0.0 ? wU : 0.0;
    double lim = 10.0 * (double)FLT_EPSILON * wU;
    lim = lim < 0.5 ? lim : 0.5;
    const double t = (diff2 <= lim) ? 0.0 : sqrt(diff2) * tnorm;
    double r;
    if (num < t) r = num / t;
    else r = 1.0;                                // both other branches give 1 for SQDIFF_NORMED (num >= 0)
    return (float)r;
}

// OpenCV templmatch.cpp common_matchTemplate(), TM_CCOEFF_NORMED branch (numType == 1, isNormed), one position.
// corr_u: sum T*I, wS1 / wU: sum I, sum I^2 over the window.  (BASELINE.json names this method; the reference calls
// TM_SQDIFF_NORMED.  The caller takes the first arg-MAX.)
// cv2's flat-window test of the TM_CCOEFF_NORMED branch, ONE place for it: `diff2` (the window's variance sum, clamped at 0)
// and whether cv2 takes the window for flat (t = 0: the result is 0 whatever sum T*I is).  finish_ccoeff_normed and
// ccoeff_ignores_corr both call this, so the decision is bit-identical whatever the compiler contracts.
__device__ __forceinline__ bool ccoeff_window_flat(double wS1, double wU, int M, double* diff2_out) {
    const double invArea = 1.0 / (double)M;
    double wndMean2 = wS1 * wS1;
    wndMean2 *= invArea;
    double diff2 = wU - wndMean2;
    diff2 = diff2 > 0.0 ? diff2 : 0.0;
    double lim = 10.0 * (double)FLT_EPSILON * wU;
    lim = lim < 0.5 ? lim : 0.5;
    *diff2_out = diff2;
    return diff2 <= lim;
}

__device__ __forceinline__ float finish_ccoeff_normed(double corr_u, double wS1, double wU, const TemplStats& ts, int M) {
    if (ts.flat) return 1.0f;
    double num = (double)(float)corr_u;          // cv2 keeps corr in its float32 result Mat
    num -= wS1 * ts.tmean;
    double diff2;
    const bool flat = ccoeff_window_flat(wS1, wU, M, &diff2);
    const double t = flat ? 0.0 : sqrt(diff2) * ts.tnorm_c;
    double r;
    if (fabs(num) < t) r = num / t;
    else if (fabs(num) < t * 1.125) r = num > 0.0 ? 1.0 : -1.0;
    else r = 0.0;
    return (float)r;
}

// Whether finish_ccoeff_normed ignores the cross term at this window: a flat pattern, or a window cv2 takes for flat (its
// variance sum inside 10 FLT_EPSILON of its energy: t = 0, the result 0 whatever sum T*I is) -- stream padding, digital silence.
// The exact stages then need not form the cross term at all.
__device__ __forceinline__ bool ccoeff_ignores_corr(double wS1, double wU, const TemplStats& ts, int M) {
    if (ts.flat) return true;
    double diff2;
    return ccoeff_window_flat(wS1, wU, M, &diff2);
}

// Score of one position from the cross term of the CENTRED samples xc = x - centre (what the direct MFMA
// kernel accumulates) and the float64 prefix sums (w1, w2: the destination stream's, offset to the window):
// sum T*I = sum T'I' + centre * (sum T + sum I) - centre^2 * M.  Returns the float32 cv2 would store at result[0][p].
__device__ __forceinline__ float score_at(double corr_c, const TemplStats& t, double centre,
                                          const double* __restrict__ w1, const double* __restrict__ w2,
                                          int64_t p, int M) {
    const double wS1 = w1[p + M] - w1[p];
    const double wU = w2[p + M] - w2[p];                             // sum I^2 over the window
    const double corr_u = corr_c + centre * (t.tS1 + wS1) - t.cM;    // sum T*I
    return finish_sqdiff_normed(corr_u, wU, t.tU, t.tnorm);
}
__device__ __forceinline__ float score_ccoeff_at(double corr_c, const TemplStats& t, double centre,
                                                 const double* __restrict__ w1, const double* __restrict__ w2,
                                                 int64_t p, int M) {
    const double wS1 = w1[p + M] - w1[p];
    const double wU = w2[p + M] - w2[p];
    const double corr_u = corr_c + centre * (t.tS1 + wS1) - t.cM;
    return finish_ccoeff_normed(corr_u, wS1, wU, t, M);
}

// Score of one position from the exact cross term of the samples themselves (refine_kernel).
__device__ __forceinline__ float score_exact(double corr_u, const TemplStats& t, const double* __restrict__ w2,
                                             int64_t p, int M) {
    return finish_sqdiff_normed(corr_u, w2[p + M] - w2[p], t.tU, t.tnorm);
}

// Overlap-save layout of one search (DESIGN.md "FFT path"): its block pairs sit on the ABSOLUTE pair grid (pair I
// starts at block FFT_STEP * I), from the pair holding the window's first position to the one holding its last.
struct FftLayout { int64_t pair0; int n_pairs; int n_seg; };
__host__ __device__ inline FftLayout fft_layout(int64_t win_start, int n_pos, int tmpl_len) {
    FftLayout l;
    l.pair0 = (win_start / FFT_SEG) / FFT_STEP;
    const int64_t pair_last = ((win_start + n_pos - 1) / FFT_SEG) / FFT_STEP;
    l.n_pairs = (int)(pair_last - l.pair0 + 1);
    l.n_seg = (tmpl_len + FFT_SEG - 1) / FFT_SEG;
    return l;
}

// segment-count class of a search: the smallest SMAX (a multiple of FFT_STEP) that holds the whole pattern.  Classes
// 0 .. MAC_SHORT_CLASSES-1 (up to 18 segments) run in mac_kernel, the others (up to 30: a 5 s pattern at 24 kHz, BASELINE
// configs[4]'s longest) in mac_long_kernel, which keeps more pattern spectra per lane at a lower occupancy; still longer
// patterns use the largest class and several chunks of its SMAX segments, the output accumulating.  (A sixth class of 36
// segments made mac_long_kernel spill at its 256 registers: 144 of them were pattern spectra.)
constexpr int MAC_CLASSES = 5;
constexpr int MAC_SHORT_CLASSES = 3;
__host__ __device__ constexpr int mac_class_smax(int c) { return FFT_STEP * (c + 1); }
__host__ __device__ inline int mac_class(int n_seg) {
    for (int c = 0; c < MAC_CLASSES - 1; ++c)
        if (n_seg <= mac_class_smax(c)) return c;
    return MAC_CLASSES - 1;
}

}  // namespace sushi
#endif
