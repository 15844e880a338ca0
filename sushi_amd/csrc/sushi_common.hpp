// sushi_amd/csrc/sushi_common.hpp -- device helpers shared by the direct (MFMA) and FFT paths.
#ifndef SUSHI_COMMON_HPP
#define SUSHI_COMMON_HPP

#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "../../include/sushi_hip.h"

namespace sushi {

constexpr int FFT_HOP = 4096;            // result positions per overlap-save block (= fft_core N / 2)
constexpr int FFT_CAND = 8;              // candidate slots per block pair (+1 truncation marker)
// Patterns shorter than this are finished by the fallback kernel (every position evaluated): the float32 error of an FFT'd cross term is
// relative to |T| * |the whole 2*FFT_HOP-sample block|, not to |T| * |the window|, so for M samples it is
// ~sqrt(2 * FFT_HOP / M) times larger in score units than the `delta` margin was measured for.
constexpr int FFT_MIN_TMPL = 2048;
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

// Template statistics in the order cv2 derives them (templmatch.cpp common_matchTemplate:
// meanStdDev -> templSum2 / templNorm), from the float64 prefix sums of the source stream's samples
// (s1 = sum x, s2 = sum x^2 of the samples as they are: exact for uint8, correctly rounded for float32).
struct TemplStats {
    double tS1;           // sum T
    double cM;            // centre^2 * M (what the direct kernel's centred cross term has to be corrected by)
    double tU, tnorm;     // templSum2, templNorm as cv2 has them
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

// Score of one position from the exact cross term of the samples themselves (refine_kernel).
__device__ __forceinline__ float score_exact(double corr_u, const TemplStats& t, const double* __restrict__ w2,
                                             int64_t p, int M) {
    return finish_sqdiff_normed(corr_u, w2[p + M] - w2[p], t.tU, t.tnorm);
}

// Overlap-save layout of one search (DESIGN.md "FFT path"); the host twin is sushi_hip_fft_layout().
struct FftLayout { int64_t k0; int n_pairs; int n_seg; };
__host__ __device__ inline FftLayout fft_layout(int64_t win_start, int n_pos, int tmpl_len) {
    FftLayout l;
    l.k0 = win_start / FFT_HOP;
    const int64_t kl = (win_start + n_pos - 1) / FFT_HOP;
    l.n_pairs = (int)((kl - l.k0 + 2) / 2);
    l.n_seg = (tmpl_len + FFT_HOP - 1) / FFT_HOP;
    return l;
}

}  // namespace sushi
#endif
