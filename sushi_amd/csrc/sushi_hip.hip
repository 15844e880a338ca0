// sushi_amd/csrc/sushi_hip.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI for Sushi's audio template match.
//
// Replaces, for a whole batch of (pattern, window) pairs, what the reference does per call in
//   wav.py:185  result = cv2.matchTemplate(search_source, pattern, cv2.TM_SQDIFF_NORMED)
//   wav.py:186  min_idx = result.argmin(axis=1)[0]
// i.e. R[p] = sum_m (T[m]-I[p+m])^2 / sqrt(sum T^2 * sum_m I[p+m]^2) with OpenCV's clamp, then first argmin.
//
// Formulation (DESIGN.md "Kernel K1"):
//   * streams are stored centred (xc = x - c, c = 0.5 | 128) so the cross term is small and the
//     sum-of-squares identity  sum (T-I)^2 = sum T'^2 - 2 sum T'I' + sum I'^2  loses nothing;
//     sum T'^2, sum I'^2, sum T', sum I' come from float64 prefix arrays built once per stream.
//   * the sliding dot product corr[p] = sum_m T'[m] I'[p+m] is computed as a GEMM with one
//     Toeplitz operand, on the exact-f32 matrix pipe (v_mfma_f32_32x32x2_f32):
//         p = base + 32 i + j ,   D[i][j] += sum_n A[i][n] B[n][j]
//         A[i][n] = T'[n - 32 i]  (zero outside [0,M))      B[n][j] = I'[base + j + n]
//     One MFMA tile therefore owns 1024 consecutive positions.  A is read from an LDS copy of
//     the template chunk laid out with a +1 skew every 32 floats (lane stride 33 -> no bank
//     conflict), B from a plain contiguous LDS copy of the search tile (lane stride 1).
//   * f32 accumulation is restarted every FLUSH template samples and folded into float64
//     accumulators, so the error of the f32 chains stays below cv2's own float32 quantum of corr.
//   * epilogue: OpenCV common_matchTemplate() in float64, result rounded to float32, packed with
//     the position into a 64-bit key; wave shuffles + LDS + one atomicMin per workgroup give the
//     first-index argmin (NumPy argmin semantics).
//
// gfx950 only: wave64, 4 SIMDs/CU, 160 KiB LDS/CU.  No CUDA compatibility paths.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "../../include/sushi_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 512;          // template samples per LDS chunk
constexpr int FLUSH = 128;       // length of one f32 accumulation chain before it is folded into float64
constexpr int ROWSPAN = 32 * 31; // 992: largest row shift 32*i of the Toeplitz operand
constexpr int TLEN = KC + ROWSPAN;              // template samples staged per chunk
constexpr int TLDS = TLEN + TLEN / 32 + 1;      // with the +1-per-32 skew

struct MatchArgs {
    const float* dst_xc;
    const double* dst_s1;
    const double* dst_s2;
    int64_t dst_len;
    const float* src_xc;
    const double* src_s1;
    const double* src_s2;
    int64_t src_len;
    double centre;
    const SushiHipSearch* searches;
    int n_search;
    int n_tiles;
    unsigned long long* keys;
};

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

// OpenCV templmatch.cpp common_matchTemplate(), TM_SQDIFF_NORMED branch, one position.
// corr_u: sum T*I (uncentred), wU: sum I^2 over the window, tU: sum T^2, tnorm: sqrt(tU).
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

template <int WAVES, int NB>
__global__ __launch_bounds__(WAVES * 64, 2)
void match_sqdiff_f32_kernel(MatchArgs a) {
    constexpr int NT = WAVES * 64;
    constexpr int TP = WAVES * NB * 1024;        // positions per workgroup
    constexpr int ILEN = TP + KC - 984;          // search samples staged per chunk: TP-1024+32 columns + KC rows + align slack, multiple of 4
    __shared__ __attribute__((aligned(16))) float lds[ILEN + TLDS];
    __shared__ unsigned long long red[WAVES];
    float* I_lds = lds;
    float* T_lds = lds + ILEN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31;                     // MFMA row (A) / column (B) index of this lane
    const int h = lane >> 5;                     // MFMA k index of this lane

    // ---- which search / which tile ------------------------------------------------------
    const int tile = xcd_remap(blockIdx.x, a.n_tiles);
    int lo = 0, hi = a.n_search - 1;             // last search with first_tile <= tile
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.searches[mid].first_tile <= tile) lo = mid; else hi = mid - 1;
    }
    const SushiHipSearch sd = a.searches[lo];
    const int s_idx = lo;
    const int M = sd.tmpl_len;
    const int P = sd.n_pos;
    const int p0 = (tile - sd.first_tile) * TP;  // first position of this workgroup
    const int wb = wave * (NB * 1024);           // first position of this wave inside the tile
    const bool wave_active = (p0 + wb) < P;

    const float* __restrict__ src = a.src_xc + sd.tmpl_off;
    const int64_t gwin = sd.win_start + p0;      // dst sample under position p0, template sample 0

    f32x16 acc[NB];
    double acc2[NB][16];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[b][r] = 0.f; acc2[b][r] = 0.0; }
    }

    const int nchunks = (M + ROWSPAN + KC - 1) / KC;
    for (int c = 0; c < nchunks; ++c) {
        const int n1 = c * KC;
        // ---- stage the search tile: dst[gwin + n1 .. + TP + KC) as aligned float4 ---------
        const int64_t g = gwin + n1;
        const int64_t gA = g & ~(int64_t)3;
        const int ioff = (int)(g - gA);
        __syncthreads();                          // previous chunk's LDS reads are done
        for (int v = tid; v < ILEN / 4; v += NT) {
            const int64_t e = gA + 4 * (int64_t)v;
            float4 val;
            if (e + 3 < a.dst_len) {
                val = *reinterpret_cast<const float4*>(a.dst_xc + e);
            } else {
                val.x = (e + 0 < a.dst_len) ? a.dst_xc[e + 0] : 0.f;
                val.y = (e + 1 < a.dst_len) ? a.dst_xc[e + 1] : 0.f;
                val.z = (e + 2 < a.dst_len) ? a.dst_xc[e + 2] : 0.f;
                val.w = 0.f;
            }
            *reinterpret_cast<float4*>(I_lds + 4 * v) = val;
        }
        // ---- stage the template chunk T'[n1-992 .. n1+KC), zero outside [0,M), skewed ------
        for (int y = tid; y < TLEN; y += NT) {
            const int x = n1 - ROWSPAN + y;
            const float v = (x >= 0 && x < M) ? src[x] : 0.f;
            T_lds[y + (y >> 5)] = v;
        }
        __syncthreads();

        if (wave_active) {
            const float* tp = T_lds + (h + 33 * (31 - i));
            const float* ip = I_lds + (ioff + wb + i + h);
            for (int nf = 0; nf < KC; nf += FLUSH) {
                // FLUSH/4 groups of two k-steps (= 4 template samples, 2*NB MFMAs).  The operands of
                // group g+1 are read from LDS before the MFMAs of group g are issued (register
                // double buffer); sched_group_barrier pins that order so the matrix pipe never waits
                // on an LDS round trip.
                const float* tq = tp + nf + (nf >> 5);
                const float* iq = ip + nf;
                float a_cur[2], b_cur[NB][2], a_nxt[2], b_nxt[NB][2];
                a_cur[0] = tq[0]; a_cur[1] = tq[2];
#pragma unroll
                for (int b = 0; b < NB; ++b) { b_cur[b][0] = iq[1024 * b]; b_cur[b][1] = iq[1024 * b + 2]; }
#pragma unroll
                for (int g = 0; g < FLUSH / 4; ++g) {
                    if (g + 1 < FLUSH / 4) {
                        const int n = 4 * (g + 1);                   // offset inside the flush block
                        const int tn = n + (n >> 5);                 // skewed template offset (nf % 32 == 0)
                        a_nxt[0] = tq[tn]; a_nxt[1] = tq[tn + 2];
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            b_nxt[b][0] = iq[n + 1024 * b]; b_nxt[b][1] = iq[n + 1024 * b + 2];
                        }
                        __builtin_amdgcn_sched_group_barrier(0x100, NB + 1, 0);   // DS reads of group g+1
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
#pragma unroll
                        for (int b = 0; b < NB; ++b)
                            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[k], b_cur[b][k], acc[b], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x8, 2 * NB, 0);         // MFMAs of group g
                    if (g + 1 < FLUSH / 4) {
                        a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1];
#pragma unroll
                        for (int b = 0; b < NB; ++b) { b_cur[b][0] = b_nxt[b][0]; b_cur[b][1] = b_nxt[b][1]; }
                    }
                }
                // fold the f32 chain (FLUSH products long) into the float64 accumulators
#pragma unroll
                for (int b = 0; b < NB; ++b) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc2[b][r] += (double)acc[b][r]; acc[b][r] = 0.f; }
                }
            }
        }
    }

    // ---- epilogue: normalise, pack (score, position), arg-min ------------------------------
    unsigned long long best = ~0ull;
    if (wave_active) {
        const double cM = a.centre * a.centre * (double)M;
        const double tS1 = a.src_s1[sd.tmpl_off + M] - a.src_s1[sd.tmpl_off];
        const double tS2 = a.src_s2[sd.tmpl_off + M] - a.src_s2[sd.tmpl_off];
        // template statistics in the order cv2 derives them (meanStdDev -> templSum2 / templNorm)
        const double t_sum = tS1 + a.centre * (double)M;          // sum T   (uncentred)
        const double t_sq = tS2 + 2.0 * a.centre * tS1 + cM;      // sum T^2 (uncentred)
        const double invArea = 1.0 / (double)M;
        const double t_mean = t_sum * invArea;
        double t_var = t_sq * invArea - t_mean * t_mean;
        t_var = t_var > 0.0 ? t_var : 0.0;
        const double t_sdv = sqrt(t_var);
        const double t_norm2 = t_sdv * t_sdv + t_mean * t_mean;   // templSum2 before "/= invArea"
        const double tU = t_norm2 / invArea;                      // templSum2
        const double tnorm = sqrt(t_norm2) / sqrt(invArea);       // templNorm
        const double* __restrict__ w1 = a.dst_s1 + sd.win_start;
        const double* __restrict__ w2 = a.dst_s2 + sd.win_start;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // C/D layout of 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int p = p0 + wb + 1024 * b + 32 * row + i;
                if (p < P) {
                    const double wS1 = w1[p + M] - w1[p];
                    const double wS2 = w2[p + M] - w2[p];
                    const double wU = wS2 + 2.0 * a.centre * wS1 + cM;               // sum I^2 over the window
                    const double corr_u = acc2[b][r] + a.centre * (tS1 + wS1) + cM;  // sum T*I
                    const float score = finish_sqdiff_normed(corr_u, wU, tU, tnorm);
                    const unsigned long long key =
                        ((unsigned long long)__float_as_uint(score) << 32) | (unsigned)p;
                    best = key < best ? key : best;
                }
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = shfl_down_u64(best, d);
        best = o < best ? o : best;
    }
    if (lane == 0) red[wave] = best;
    __syncthreads();
    if (tid == 0) {
        unsigned long long m = red[0];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) m = red[w] < m ? red[w] : m;
        if (m != ~0ull) atomicMin(a.keys + s_idx, m);
    }
}

__global__ void unpack_keys_kernel(const unsigned long long* __restrict__ keys, int n,
                                   int32_t* __restrict__ out_idx, float* __restrict__ out_score) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) {
        const unsigned long long key = keys[k];
        out_idx[k] = (int32_t)(unsigned)(key & 0xffffffffull);
        out_score[k] = __uint_as_float((unsigned)(key >> 32));
    }
}

// ------------------------------------------------------------------------------------------
// Stream preparation: centred float32 copy + float64 exclusive prefix sums of xc and xc^2.
// Three passes over blocks of PB samples (block totals -> scan of totals -> in-block scan).
// ------------------------------------------------------------------------------------------
constexpr int PB_THREADS = 256;
constexpr int PB_PER_THREAD = 16;
constexpr int PB = PB_THREADS * PB_PER_THREAD;   // 4096 samples per block

template <typename T> __device__ __forceinline__ float centred(T x);
template <> __device__ __forceinline__ float centred<float>(float x) { return x - 0.5f; }
template <> __device__ __forceinline__ float centred<uint8_t>(uint8_t x) { return (float)((int)x - 128); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    return v;
}

template <typename T>
__global__ __launch_bounds__(PB_THREADS)
void centre_blocksum_kernel(const T* __restrict__ raw, int64_t n, float* __restrict__ xc,
                            double* __restrict__ bs1, double* __restrict__ bs2) {
    __shared__ double r1[PB_THREADS / 64], r2[PB_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * PB;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
    for (int k = 0; k < PB_PER_THREAD; ++k) {
        const int64_t e = base + (int64_t)k * PB_THREADS + threadIdx.x;   // coalesced
        if (e < n) {
            const float v = centred<T>(raw[e]);
            xc[e] = v;
            s1 += (double)v;
            s2 += (double)v * (double)v;
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t1 = 0.0, t2 = 0.0;
        for (int w = 0; w < PB_THREADS / 64; ++w) { t1 += r1[w]; t2 += r2[w]; }
        bs1[blockIdx.x] = t1;
        bs2[blockIdx.x] = t2;
    }
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

// single workgroup: in-place exclusive scan of the per-block totals
__global__ __launch_bounds__(1024)
void scan_blocksums_kernel(double* __restrict__ bs1, double* __restrict__ bs2, int nb) {
    __shared__ double w1[16], w2[16];
    __shared__ double carry[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { carry[0] = 0.0; carry[1] = 0.0; }
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int k = base + tid;
        const double v1 = k < nb ? bs1[k] : 0.0;
        const double v2 = k < nb ? bs2[k] : 0.0;
        double t1, t2;
        const double e1 = wave_excl_scan(v1, &t1);
        const double e2 = wave_excl_scan(v2, &t2);
        if (lane == 0) { w1[wv] = t1; w2[wv] = t2; }
        __syncthreads();
        double o1 = carry[0], o2 = carry[1];
        for (int w = 0; w < wv; ++w) { o1 += w1[w]; o2 += w2[w]; }
        if (k < nb) { bs1[k] = o1 + e1; bs2[k] = o2 + e2; }
        __syncthreads();
        if (tid == 1023) { carry[0] = o1 + e1 + v1; carry[1] = o2 + e2 + v2; }
        __syncthreads();
    }
}

__global__ __launch_bounds__(PB_THREADS)
void final_scan_kernel(const float* __restrict__ xc, int64_t n, const double* __restrict__ bs1,
                       const double* __restrict__ bs2, double* __restrict__ s1, double* __restrict__ s2) {
    __shared__ double w1[PB_THREADS / 64], w2[PB_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * PB + (int64_t)tid * PB_PER_THREAD;  // 16 consecutive samples
    float v[PB_PER_THREAD];
    double l1 = 0.0, l2 = 0.0;
#pragma unroll
    for (int k = 0; k < PB_PER_THREAD; ++k) {
        const int64_t e = base + k;
        v[k] = e < n ? xc[e] : 0.f;
        l1 += (double)v[k];
        l2 += (double)v[k] * (double)v[k];
    }
    double t1, t2;
    double e1 = wave_excl_scan(l1, &t1);
    double e2 = wave_excl_scan(l2, &t2);
    if (lane == 0) { w1[wv] = t1; w2[wv] = t2; }
    __syncthreads();
    double o1 = bs1[blockIdx.x], o2 = bs2[blockIdx.x];
    for (int w = 0; w < wv; ++w) { o1 += w1[w]; o2 += w2[w]; }
    double r1 = o1 + e1, r2 = o2 + e2;           // exclusive prefix at sample `base`
    if (blockIdx.x == 0 && tid == 0) { s1[0] = 0.0; s2[0] = 0.0; }
#pragma unroll
    for (int k = 0; k < PB_PER_THREAD; ++k) {
        const int64_t e = base + k;
        if (e < n) {
            r1 += (double)v[k];
            r2 += (double)v[k] * (double)v[k];
            s1[e + 1] = r1;
            s2[e + 1] = r2;
        }
    }
}

inline int launch_ok() { return hipGetLastError() == hipSuccess ? SUSHI_HIP_OK : SUSHI_HIP_ELAUNCH; }

struct Variant { int waves, nb; };
constexpr Variant kVariants[] = {{1, 1}, {4, 1}, {4, 4}};
constexpr int kNumVariants = 3;

}  // namespace

extern "C" {

int sushi_hip_abi_version(void) { return SUSHI_HIP_ABI_VERSION; }

const char* sushi_hip_strerror(int code) {
    switch (code) {
        case SUSHI_HIP_OK: return "ok";
        case SUSHI_HIP_EINVAL: return "invalid argument";
        case SUSHI_HIP_EALIGN: return "device pointer not aligned";
        case SUSHI_HIP_ELAUNCH: return "HIP launch failed";
        case SUSHI_HIP_ENOSPACE: return "workspace too small";
        case SUSHI_HIP_ENODEV: return "no gfx950 device";
        default: return "unknown sushi_hip error";
    }
}

int sushi_hip_device_ok(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SUSHI_HIP_ENODEV;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return SUSHI_HIP_ENODEV;
    const char* arch = prop.gcnArchName;
    // "gfx950:sramecc+:xnack-"
    if (arch[0] == 'g' && arch[1] == 'f' && arch[2] == 'x' && arch[3] == '9' && arch[4] == '5' && arch[5] == '0')
        return SUSHI_HIP_OK;
    return SUSHI_HIP_ENODEV;
}

int sushi_hip_variant_count(void) { return kNumVariants; }

int sushi_hip_variant_tile_positions(int variant) {
    if (variant < 0 || variant >= kNumVariants) return SUSHI_HIP_EINVAL;
    return kVariants[variant].waves * kVariants[variant].nb * 1024;
}

double sushi_hip_centre(int dtype) { return dtype == SUSHI_HIP_U8 ? 128.0 : 0.5; }

size_t sushi_hip_prepare_workspace_bytes(int64_t n) {
    if (n < 0) return 0;
    const int64_t nb = (n + PB - 1) / PB;
    return (size_t)(2 * (nb > 0 ? nb : 1)) * sizeof(double);
}

int sushi_hip_prepare_stream(const void* raw_dev, int dtype, int64_t n, float* xc_dev, double* s1_dev,
                             double* s2_dev, void* ws_dev, size_t ws_bytes, void* hip_stream) {
    if (!raw_dev || !xc_dev || !s1_dev || !s2_dev || !ws_dev || n <= 0) return SUSHI_HIP_EINVAL;
    if (dtype != SUSHI_HIP_U8 && dtype != SUSHI_HIP_F32) return SUSHI_HIP_EINVAL;
    if (((uintptr_t)xc_dev & 15) || ((uintptr_t)s1_dev & 7) || ((uintptr_t)s2_dev & 7) || ((uintptr_t)ws_dev & 7))
        return SUSHI_HIP_EALIGN;
    if (ws_bytes < sushi_hip_prepare_workspace_bytes(n)) return SUSHI_HIP_ENOSPACE;
    const int64_t nb64 = (n + PB - 1) / PB;
    if (nb64 > 0x7fffffff) return SUSHI_HIP_EINVAL;
    const int nb = (int)nb64;
    hipStream_t st = (hipStream_t)hip_stream;
    double* bs1 = (double*)ws_dev;
    double* bs2 = bs1 + nb;
    if (dtype == SUSHI_HIP_F32)
        hipLaunchKernelGGL(centre_blocksum_kernel<float>, dim3(nb), dim3(PB_THREADS), 0, st,
                           (const float*)raw_dev, n, xc_dev, bs1, bs2);
    else
        hipLaunchKernelGGL(centre_blocksum_kernel<uint8_t>, dim3(nb), dim3(PB_THREADS), 0, st,
                           (const uint8_t*)raw_dev, n, xc_dev, bs1, bs2);
    if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
    hipLaunchKernelGGL(scan_blocksums_kernel, dim3(1), dim3(1024), 0, st, bs1, bs2, nb);
    if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
    hipLaunchKernelGGL(final_scan_kernel, dim3(nb), dim3(PB_THREADS), 0, st, (const float*)xc_dev, n,
                       (const double*)bs1, (const double*)bs2, s1_dev, s2_dev);
    return launch_ok();
}

int sushi_hip_match_batch(const float* dst_xc_dev, const double* dst_s1_dev, const double* dst_s2_dev, int64_t dst_len,
                          const float* src_xc_dev, const double* src_s1_dev, const double* src_s2_dev, int64_t src_len,
                          double centre, int method, const SushiHipSearch* searches_dev, int n_search, int n_tiles,
                          int variant, uint64_t* keys_ws_dev, int32_t* out_idx_dev, float* out_score_dev,
                          void* hip_stream) {
    if (!dst_xc_dev || !dst_s1_dev || !dst_s2_dev || !src_xc_dev || !src_s1_dev || !src_s2_dev || !searches_dev ||
        !keys_ws_dev || !out_idx_dev || !out_score_dev)
        return SUSHI_HIP_EINVAL;
    if (dst_len <= 0 || src_len <= 0 || n_search <= 0 || n_tiles < n_search) return SUSHI_HIP_EINVAL;
    if (method != SUSHI_HIP_SQDIFF_NORMED) return SUSHI_HIP_EINVAL;
    if (variant < 0 || variant >= kNumVariants) return SUSHI_HIP_EINVAL;
    if (((uintptr_t)dst_xc_dev & 15) || ((uintptr_t)keys_ws_dev & 7) || ((uintptr_t)searches_dev & 7))
        return SUSHI_HIP_EALIGN;
    hipStream_t st = (hipStream_t)hip_stream;
    if (hipMemsetAsync(keys_ws_dev, 0xff, (size_t)n_search * sizeof(uint64_t), st) != hipSuccess)
        return SUSHI_HIP_ELAUNCH;
    MatchArgs a;
    a.dst_xc = dst_xc_dev; a.dst_s1 = dst_s1_dev; a.dst_s2 = dst_s2_dev; a.dst_len = dst_len;
    a.src_xc = src_xc_dev; a.src_s1 = src_s1_dev; a.src_s2 = src_s2_dev; a.src_len = src_len;
    a.centre = centre; a.searches = searches_dev; a.n_search = n_search; a.n_tiles = n_tiles;
    a.keys = (unsigned long long*)keys_ws_dev;
    switch (variant) {
        case 0: hipLaunchKernelGGL((match_sqdiff_f32_kernel<1, 1>), dim3(n_tiles), dim3(64), 0, st, a); break;
        case 1: hipLaunchKernelGGL((match_sqdiff_f32_kernel<4, 1>), dim3(n_tiles), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((match_sqdiff_f32_kernel<4, 4>), dim3(n_tiles), dim3(256), 0, st, a); break;
    }
    if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
    hipLaunchKernelGGL(unpack_keys_kernel, dim3((n_search + 255) / 256), dim3(256), 0, st,
                       (const unsigned long long*)keys_ws_dev, n_search, out_idx_dev, out_score_dev);
    return launch_ok();
}

}  // extern "C"
