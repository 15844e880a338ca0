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
#include <algorithm>
#include <stdint.h>
#include <float.h>

#include <new>

#include "../../include/sushi_hip.h"
#include "sushi_common.hpp"
#include "sushi_internal.hpp"

namespace {

using namespace sushi;

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
    const SearchDesc* searches;
    int n_search;
    int n_tiles;
    int method;                   // SUSHI_HIP_METHOD_*
    unsigned long long* keys;
};

template <int WAVES, int NB> struct TileShape {
    static constexpr int NT = WAVES * 64;
    static constexpr int TP = WAVES * NB * 1024;        // positions per workgroup
    static constexpr int ILEN = TP + KC - 984;          // search samples staged per chunk: TP-1024+32 columns + KC rows + align slack, multiple of 4
    static constexpr int LDS_FLOATS = ILEN + TLDS;
};

// One tile (TP consecutive result positions) of one search.  `lds` holds LDS_FLOATS floats, `red` WAVES keys.
template <int WAVES, int NB>
__device__ __forceinline__ void match_tile(const MatchArgs& a, const int s_idx, const SearchDesc sd,
                                           const int tile_in_search, float* lds, unsigned long long* red) {
    constexpr int NT = TileShape<WAVES, NB>::NT;
    constexpr int TP = TileShape<WAVES, NB>::TP;
    constexpr int ILEN = TileShape<WAVES, NB>::ILEN;
    float* I_lds = lds;
    float* T_lds = lds + ILEN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31;                     // MFMA row (A) / column (B) index of this lane
    const int h = lane >> 5;                     // MFMA k index of this lane

    const int M = sd.tmpl_len;
    const int P = sd.n_pos;
    const int p0 = tile_in_search * TP;          // first position of this workgroup
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
        const TemplStats ts = templ_stats(a.src_s1, a.src_s2, sd.tmpl_off, M, a.centre);
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
                    const unsigned long long key = a.method == SUSHI_HIP_METHOD_CCOEFF_NORMED
                        ? make_key_max(score_ccoeff_at(acc2[b][r], ts, a.centre, w1, w2, p, M), (unsigned)p)
                        : make_key(score_at(acc2[b][r], ts, a.centre, w1, w2, p, M), (unsigned)p);
                    best = key < best ? key : best;
                }
            }
        }
    }
    best = wave_min_u64(best);
    if (lane == 0) red[wave] = best;
    __syncthreads();
    if (tid == 0) {
        unsigned long long m = red[0];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) m = red[w] < m ? red[w] : m;
        if (m != NO_KEY) atomicMin(a.keys + s_idx, m);
    }
}


template <int WAVES, int NB>
__global__ __launch_bounds__(WAVES * 64, 2)
void match_sqdiff_f32_kernel(MatchArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[TileShape<WAVES, NB>::LDS_FLOATS];
    __shared__ unsigned long long red[WAVES];
    // ---- which search / which tile ------------------------------------------------------
    const int tile = xcd_remap(blockIdx.x, a.n_tiles);
    int lo = 0, hi = a.n_search - 1;             // last search with first_tile <= tile
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.searches[mid].first_tile <= tile) lo = mid; else hi = mid - 1;
    }
    const SearchDesc sd = a.searches[lo];
    match_tile<WAVES, NB>(a, lo, sd, tile - sd.first_tile, lds, red);
}

// ------------------------------------------------------------------------------------------
// FFT path, exact stages.  One accumulation order everywhere (refine_kernel, both modes of exact_tiles_kernel):
// sum T*I over the samples as they are, in float64 (every product of two float32 or uint8 values is exact there),
// XM pattern samples at a time -- each chunk summed sequentially from 0 with one fused multiply-add per sample,
// the chunk sums added in order.  The value of a position therefore does not depend on which kernel, which tile
// or which window evaluated it.
// ------------------------------------------------------------------------------------------
constexpr int XT = TILE;         // positions per tile
constexpr int XM = 512;          // pattern samples per chunk of the canonical sum
static_assert(XT == 1024 && XM % 4 == 0, "exact_tiles_kernel: 256 threads x 4 consecutive positions, steps of 4 samples");

// 16 bytes of a stream at any element boundary as ONE load: a thread walks its own chunk, so neighbouring lanes read 2 KB
// apart and every load instruction touches 64 different lines whatever its width -- a dwordx4 brings four (sixteen) samples
// for the price of one (gfx950 global loads need no more than byte alignment; the order of the additions is untouched)
template <typename T> struct WideLoad;
template <> struct WideLoad<float> { static constexpr int N = 4; struct __attribute__((packed, aligned(4))) V { float v[4]; }; };
template <> struct WideLoad<uint8_t> { static constexpr int N = 16; struct __attribute__((packed, aligned(1))) V { uint8_t v[16]; }; };

// NPOS consecutive positions of one search over one chunk of the pattern (a candidate is the first of four; an audit run is four,
// the four runs of one pair sixteen): position q sees the window w + q, so a step's window samples serve all of them -- per step
// K pattern loads and K window loads (the window vectors behind the step's own are the first of the next step's).  Every
// position keeps its own chain in the canonical order -- one fused multiply-add per sample, samples in order -- so its value
// does not depend on what it was evaluated with.  A thread walks its own chunk: neighbouring lanes read 2 KB apart, every
// load instruction touches 64 lines whatever its width; the loads of step i + 1 are issued before the additions of step i.
// room: window elements readable from w on.
template <typename T, int NPOS>
__device__ __forceinline__ void chunk_run(const T* __restrict__ t, const T* __restrict__ w, const int mc, const int64_t room,
                                          double (&acc)[NPOS]) {
    typedef typename WideLoad<T>::V V;
    constexpr int N = WideLoad<T>::N;
    constexpr int K = N >= 16 ? 1 : 2;                   // a step = 16 uint8 / 8 float32 samples
    constexpr int STEP = K * N;
    constexpr int OV = (NPOS - 1 + N - 1) / N;           // vectors a step's positions reach past the step's own samples
#pragma unroll
    for (int q = 0; q < NPOS; ++q) acc[q] = 0.0;
    // whole steps whose window loads (this step's K + OV vectors, the next step's K) stay inside `room`
    const int64_t fit = room >= (int64_t)(STEP + N * OV) ? (room - N * OV) / STEP : 0;
    const int steps = fit < (int64_t)(mc / STEP) ? (int)fit : mc / STEP;
    V a[K], an[K], b[K + OV], bn[K];
    if (steps > 0) {
#pragma unroll
        for (int j = 0; j < K; ++j) a[j] = *reinterpret_cast<const V*>(t + N * j);
#pragma unroll
        for (int j = 0; j < K + OV; ++j) b[j] = *reinterpret_cast<const V*>(w + N * j);
    }
    for (int s = 0; s < steps; ++s) {
        const int mn = (s + 1 < steps ? s + 1 : s) * STEP;          // (the last step re-requests itself: unconditional loads)
#pragma unroll
        for (int j = 0; j < K; ++j) an[j] = *reinterpret_cast<const V*>(t + mn + N * j);
#pragma unroll
        for (int j = 0; j < K; ++j) bn[j] = *reinterpret_cast<const V*>(w + mn + N * (OV + j));
#pragma unroll
        for (int e = 0; e < STEP; ++e) {
            const double te = (double)a[e / N].v[e % N];
#pragma unroll
            for (int q = 0; q < NPOS; ++q) acc[q] = __builtin_fma(te, (double)b[(e + q) / N].v[(e + q) % N], acc[q]);
        }
#pragma unroll
        for (int j = 0; j < K; ++j) a[j] = an[j];
#pragma unroll
        for (int j = 0; j < OV; ++j) b[j] = b[j + K];
#pragma unroll
        for (int j = 0; j < K; ++j) b[OV + j] = bn[j];
    }
    for (int m = steps * STEP; m < mc; ++m) {
        const double te = (double)t[m];
#pragma unroll
        for (int q = 0; q < NPOS; ++q) {
            const int64_t x = (int64_t)m + q < room ? (int64_t)m + q : room - 1;       // (past the stream: a position nobody reads)
            acc[q] = __builtin_fma(te, (double)w[x], acc[q]);
        }
    }
}

// Exact evaluation of the tiles collect_kernel listed: every valid position of a dense tile (a thread owns four
// CONSECUTIVE positions: the pattern samples T[m .. m+3] then meet the seven search samples I[p .. p+6], of which
// four are the previous step's -- one 16-byte LDS read of each per 16 float64 FMAs), or the listed candidate
// positions of a sparse tile.  XG chunks of the pattern are staged at a time.  Fixed grid, workgroups stride over the
// tile list; with no tile listed every workgroup leaves after one load.
//
// Runs of EQUAL samples (digital silence: wav.py:148-151 maps it to one mid-level value, and a pattern cut from such a span
// ties over the whole run -- dense tiles inside or at an end of such a run are most of what a tie-saturated job lists).  Where
// every sample the tile's positions read in a chunk is the same value v, every position's sum over that chunk is the same
// chain of multiply-adds fma(T[m], v, .): it is formed ONCE per chunk (a thread takes a chunk) and added to every position's
// total where the loop would have added the position's own -- the same operations in the same order, so the same bits.
// A tile inside a run costs its chunk chains; a tile at an end of one, the two or three chunks that reach outside it.
//
// A sparse tile's candidates each walk a chain of M dependent multiply-adds: with one thread per candidate a tile of 30
// candidates (a held tone: one every period) kept one wave busy for 50 chunks, 0.3 ms, and a tie-saturated job lists two
// thousand of them.  Thread <-> (staged chunk, candidate): the chunk sums go through the LDS and are added in chunk order.
constexpr int XG = 4;            // chunks staged together
constexpr int XMP = XM + 2;      // a staged chunk of the pattern + padding: lanes on different chunks read different banks
// A sparse tile's lanes read the staged window at their candidates' offsets: candidates a fixed distance apart (a held tone's
// period: 30 samples = 240 bytes) landed on 8 of the 32 banks -- 8-way conflicts on every read, 0.6 ms a tile.  One padding
// element per 32 spreads any such stride over the banks.
__device__ __forceinline__ int skew(const int e) { return e + (e >> 5); }
constexpr int XRUN = 256;        // patterns of up to XRUN chunks (131,072 samples) are looked at for runs of equal samples
template <typename T>
__global__ __launch_bounds__(256)
void exact_tiles_kernel(TileParams a) {
    constexpr int XQ = XT / 256;
    // chunks are staged as float64: the conversions (quarter rate) happen once per staged sample, not once per multiply-add
    __shared__ __attribute__((aligned(16))) double lt[XG * XMP];
    __shared__ __attribute__((aligned(16))) double li[XT + XG * XM + (XT + XG * XM) / 32 + 1];
    __shared__ double part[XG * SPARSE_TILE_MAX];     // dense: the runs' chunk chains ([n_chunks]); sparse: the staged chunks' sums per candidate
    __shared__ double ctot[SPARSE_TILE_MAX];          // sparse: the candidates' totals
    __shared__ int cpos[SPARSE_TILE_MAX];             // sparse: the candidates' positions, relative to the tile
    __shared__ unsigned long long red[4];
    __shared__ int run_edge[2], next_item;
    static_assert(XRUN <= XG * SPARSE_TILE_MAX, "the chains share the chunk sums' space");
    const int n_tiles = a.sub->n_tiles;
    if (n_tiles == 0) return;
    const int tid = threadIdx.x;
    typedef double d2 __attribute__((ext_vector_type(2)));
    // A few tiles only (a couple of flagged searches) are a latency problem, not a throughput one: a dense tile is then cut
    // into XQ workgroups of one position per thread -- the chain of dependent multiply-adds a thread walks is a quarter as long.
    // Each position's sum is the same chain either way.
    const int split = n_tiles * 2 <= (int)gridDim.x ? XQ : 1;
    // The entries cost from microseconds (a tile inside a run) to a millisecond (every position of a tile, a long pattern):
    // the workgroups take them off a queue one at a time -- handed out in equal shares, some workgroups ran twice as long as others.
    for (;;) {
        __syncthreads();                                                // the previous entry's shared state is consumed
        if (tid == 0) next_item = atomicAdd(&a.sub->tile_next, 1);
        __syncthreads();
        const int item = next_item;
        if (item >= n_tiles * split) break;
        const int v = item / split, sub = item - v * split;
        const TileDesc td = a.tiles[v];
        const SearchDesc sd = a.searches[td.search];
        const int M = sd.tmpl_len;
        const int n_chunks = (M + XM - 1) / XM;
        const bool dense = td.cnt < 0;
        if (!dense && sub > 0) continue;                                // a sparse tile is one workgroup's work (uniform)
        const bool quarter = dense && split > 1;
        const int p0 = td.p0 + (quarter ? sub * 256 : 0);               // may be negative: tiles sit on the absolute grid
        const int span = quarter ? 256 : XT;                            // positions this workgroup covers
        const int cnt = dense ? 0 : min(td.cnt, SPARSE_TILE_MAX);
        const T* __restrict__ Tp = (const T*)a.r.src_raw + sd.tmpl_off;
        const T* __restrict__ Ip = (const T*)a.r.dst_raw + (sd.win_start + p0);
        const int64_t room = a.r.dst_len - (sd.win_start + p0);         // samples of the stream from Ip on
        if (tid < cnt) { cpos[tid] = a.cand[td.off + tid] - p0; ctot[tid] = 0.0; }
        const TemplStats ts = templ_stats(a.r.src_s1, a.r.src_s2, sd.tmpl_off, M, a.r.centre);
        const double* __restrict__ w1 = a.r.dst_s1 + sd.win_start;
        const double* __restrict__ w2 = a.r.dst_s2 + sd.win_start;
        const bool ccoeff = a.method == SUSHI_HIP_METHOD_CCOEFF_NORMED;
        // TM_CCOEFF_NORMED: where cv2 takes the window for flat (stream padding, digital silence) its result is 0 whatever the
        // cross term is -- which the float64 prefix sums alone decide.  A tile whose every position is like that needs no sums
        // at all (such tiles are why a search that touches the padding is flagged in the first place).
        bool needs_corr = true;
        if (ccoeff) {
            int need = 0;
            if (quarter) {
                const int p = p0 + tid;
                need = p >= 0 && p < sd.n_pos && !ccoeff_ignores_corr(w1[p + M] - w1[p], w2[p + M] - w2[p], ts, M);
            } else if (dense) {
#pragma unroll
                for (int q = 0; q < XQ; ++q) {
                    const int p = p0 + XQ * tid + q;
                    need |= p >= 0 && p < sd.n_pos && !ccoeff_ignores_corr(w1[p + M] - w1[p], w2[p + M] - w2[p], ts, M);
                }
            } else if (tid < cnt) {
                const int p = a.cand[td.off + tid];
                need = !ccoeff_ignores_corr(w1[p + M] - w1[p], w2[p + M] - w2[p], ts, M);
            }
            needs_corr = __syncthreads_or(need) != 0;
        }
        // the runs of equal samples at the tile's first and last sample: chunks [0, na) see nothing but the first one's value,
        // chunks [nb, n_chunks) nothing but the last one's
        int na = 0, nb = n_chunks;
        if (dense && needs_corr && (int64_t)span + M - 1 <= room && n_chunks <= XRUN) {
            typedef typename WideLoad<T>::V V;
            constexpr int N = WideLoad<T>::N;
            const int need = span + M - 1;                              // samples the tile's positions read
            const T v0 = Ip[0], v1 = Ip[need - 1];
            int e1 = need, e2 = -1;                                     // first sample that differs from v0 / last that differs from v1
            int e = tid * N;
#pragma unroll 4
            for (; e + N <= need; e += 256 * N) {
                const V x = *reinterpret_cast<const V*>(Ip + e);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    if (!(x.v[j] == v0)) e1 = min(e1, e + j);
                    if (!(x.v[j] == v1)) e2 = max(e2, e + j);
                }
            }
            for (; e < need; ++e) {                                     // (one thread's tail of fewer than N samples)
                if (!(Ip[e] == v0)) e1 = min(e1, e);
                if (!(Ip[e] == v1)) e2 = max(e2, e);
            }
            if (tid == 0) { run_edge[0] = need; run_edge[1] = -1; }
            __syncthreads();
            if (e1 < need) atomicMin(&run_edge[0], e1);
            if (e2 >= 0) atomicMax(&run_edge[1], e2);
            __syncthreads();
            e1 = run_edge[0]; e2 = run_edge[1];
            // chunk c covers pattern samples [c XM, c XM + mc): the tile's positions read window samples [c XM, c XM + mc + span - 1) for it
            if (e1 >= need) na = n_chunks;
            else na = max(min((e1 - (span - 1)) / XM, n_chunks - 1), 0);      // (whole chunks only: the last one may be short, and is then not the run's)
            nb = na >= n_chunks ? n_chunks : max(e2 / XM + 1, na);      // c XM > e2
            for (int c = tid; c < n_chunks; c += 256) {
                if (c >= na && c < nb) continue;
                const double vd = c < na ? (double)v0 : (double)v1;
                const T* __restrict__ tc = Tp + (size_t)c * XM;
                const int mc = min(XM, M - c * XM);
                double acc = 0.0;
                int m = 0;
#pragma unroll 8
                for (; m + N <= mc; m += N) {                           // (a thread walks its own chunk: loads well ahead of the chain)
                    const V x = *reinterpret_cast<const V*>(tc + m);
#pragma unroll
                    for (int j = 0; j < N; ++j) acc = __builtin_fma((double)x.v[j], vd, acc);
                }
                for (; m < mc; ++m) acc = __builtin_fma((double)tc[m], vd, acc);
                part[c] = acc;
            }
        }
        double tot[XQ] = {0.0, 0.0, 0.0, 0.0};
        for (int c0 = 0; needs_corr && c0 < n_chunks; c0 += XG) {
            const int gc = min(XG, n_chunks - c0);
            const int m0 = c0 * XM;
            __syncthreads();                                            // previous group's reads are done (and part[] of the runs is written)
            const bool runs_only = c0 + gc <= na || c0 >= nb;           // (uniform)
            if (!runs_only) {
                // (staging: SB loads of a thread are requested together, from clamped addresses, and selected afterwards -- a
                // conditional load per element came one after the other, a microsecond each, and was most of a sparse tile's time)
                constexpr int SB = 8;
                auto stage = [&](const T* __restrict__ src, const int64_t valid, const int n, auto put) {
                    const int64_t last = valid > 0 ? valid - 1 : 0;
                    for (int e0 = tid; e0 < n; e0 += SB * 256) {
                        T x[SB];
#pragma unroll
                        for (int u = 0; u < SB; ++u) {
                            const int64_t g = (int64_t)m0 + e0 + 256 * u;
                            x[u] = src[g < last ? g : last];
                        }
#pragma unroll
                        for (int u = 0; u < SB; ++u) {
                            const int e = e0 + 256 * u;
                            if (e < n) put(e, (int64_t)m0 + e < valid ? (double)x[u] : 0.0);
                        }
                    }
                };
                stage(Tp, (int64_t)M, gc * XM, [&](const int e, const double x) { lt[(e / XM) * XMP + e % XM] = x; });   // zero padded: whole steps of 4
                if (dense) stage(Ip, room, span + gc * XM, [&](const int e, const double x) { li[e] = x; });
                else stage(Ip, room, span + gc * XM, [&](const int e, const double x) { li[skew(e)] = x; });
                __syncthreads();
            }
            if (dense) {
                for (int ch = 0; ch < gc; ++ch) {
                    const int c = c0 + ch;
                    if (c < na || c >= nb) {                            // the run's chain for this chunk: every position's chunk sum
                        const double r = part[c];
#pragma unroll
                        for (int q = 0; q < XQ; ++q) tot[q] += r;
                        continue;
                    }
                    const int mc = min(XM, M - c * XM);
                    const d2* __restrict__ lt2 = reinterpret_cast<const d2*>(lt + ch * XMP);
                    if (quarter) {
                        const double* __restrict__ wv = li + ch * XM + tid;
                        double acc = 0.0;
                        for (int k = 0; k < (mc + 3) / 4; ++k) {
                            const d2 ta = lt2[2 * k], tb = lt2[2 * k + 1];
                            const double x0 = wv[4 * k], x1 = wv[4 * k + 1], x2 = wv[4 * k + 2], x3 = wv[4 * k + 3];
                            acc = __builtin_fma(ta.x, x0, acc);                // pattern samples in order (padding adds exact zeros)
                            acc = __builtin_fma(ta.y, x1, acc);
                            acc = __builtin_fma(tb.x, x2, acc);
                            acc = __builtin_fma(tb.y, x3, acc);
                        }
                        tot[0] += acc;
                    } else {
                        const d2* __restrict__ li2 = reinterpret_cast<const d2*>(li + ch * XM) + 2 * tid;   // li[4 tid + 4 k ..]
                        double acc[XQ] = {0.0, 0.0, 0.0, 0.0};
                        d2 lo0 = li2[0], lo1 = li2[1];
                        for (int k = 0; k < (mc + 3) / 4; ++k) {
                            const d2 hi0 = li2[2 * k + 2], hi1 = li2[2 * k + 3];
                            const d2 ta = lt2[2 * k], tb = lt2[2 * k + 1];
                            const double w[8] = {lo0.x, lo0.y, lo1.x, lo1.y, hi0.x, hi0.y, hi1.x, hi1.y};
                            const double t[4] = {ta.x, ta.y, tb.x, tb.y};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {                      // pattern samples in order (padding adds exact zeros)
#pragma unroll
                                for (int q = 0; q < XQ; ++q) acc[q] = __builtin_fma(t[j], w[q + j], acc[q]);
                            }
                            lo0 = hi0; lo1 = hi1;
                        }
#pragma unroll
                        for (int q = 0; q < XQ; ++q) tot[q] += acc[q];
                    }
                }
            } else {
                // thread <-> (staged chunk, candidate), chunk-major: the lanes of a wave read the same pattern sample
                // (eight LDS reads of each operand in flight per step: one read, one multiply-add at a time, a listed candidate
                // took as long as a whole dense tile)
                for (int i = tid; i < gc * cnt; i += 256) {
                    const int ch = i / cnt, j = i - ch * cnt;
                    const int mc = min(XM, M - (c0 + ch) * XM);
                    const double* __restrict__ tv_ = lt + ch * XMP;
                    const int wb = ch * XM + cpos[j];                  // (the candidate's window in the skewed staging)
                    double acc = 0.0;
                    int m = 0;
                    for (; m + 8 <= mc; m += 8) {
                        double tv[8], wq[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { tv[u] = tv_[m + u]; wq[u] = li[skew(wb + m + u)]; }
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc = __builtin_fma(tv[u], wq[u], acc);
                    }
                    for (; m < mc; ++m) acc = __builtin_fma(tv_[m], li[skew(wb + m)], acc);
                    part[i] = acc;
                }
                __syncthreads();
                if (tid < cnt) {
                    double t = ctot[tid];
                    for (int ch = 0; ch < gc; ++ch) t += part[ch * cnt + tid];      // chunk sums in order
                    ctot[tid] = t;
                }
            }
        }
        auto key_at = [&](const double corr_u, const int p) {
            return ccoeff ? make_key_max(finish_ccoeff_normed(corr_u, w1[p + M] - w1[p], w2[p + M] - w2[p], ts, M), (unsigned)p)
                          : make_key(score_exact(corr_u, ts, w2, (int64_t)p, M), (unsigned)p);
        };
        unsigned long long best = NO_KEY;
        if (quarter) {
            const int p = p0 + tid;
            if (p >= 0 && p < sd.n_pos) best = key_at(tot[0], p);
        } else if (dense) {
#pragma unroll
            for (int q = 0; q < XQ; ++q) {
                const int p = p0 + XQ * tid + q;
                if (p >= 0 && p < sd.n_pos) {
                    const unsigned long long key = key_at(tot[q], p);
                    best = key < best ? key : best;
                }
            }
        } else if (tid < cnt) {
            best = key_at(ctot[tid], p0 + cpos[tid]);
        }
        best = wave_min_u64(best);
        if ((tid & 63) == 0) red[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w) best = red[w] < best ? red[w] : best;
            if (best != NO_KEY) atomicMin(a.keys + td.search, best);
        }
    }
}

__global__ void unpack_keys_kernel(const unsigned long long* __restrict__ keys, int n, int method,
                                   int32_t* __restrict__ out_idx, float* __restrict__ out_score, int2* __restrict__ out_packed) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) {
        const unsigned long long key = keys[k];
        const float sc = method == SUSHI_HIP_METHOD_CCOEFF_NORMED ? key_score_max(key) : key_score(key);
        out_idx[k] = (int32_t)key_pos(key);
        out_score[k] = sc;
        // the same pair once more as ONE 8-byte record (index, score bits): what a rank hands to the all-gather as it is
        if (out_packed) out_packed[k] = int2{(int)key_pos(key), __float_as_int(sc)};
    }
}

// ------------------------------------------------------------------------------------------
// FFT path: the candidates ifft_kernel listed per pair.  A position is a candidate of the search if its lower
// bound (f32 score - the pair's error bound) is not above the search's threshold U = the smallest (f32 score +
// bound) over all pairs: the exact minimum cannot lie anywhere else.  Up to RCAP candidates are evaluated here,
// exactly; a search with more, or with a pair that could not list all of its own, is flagged for the
// collection pass + exact_tiles_kernel (flag 1).  Every evaluated candidate also checks the bound it was selected
// with: an exact score further from the f32 one than the bound allows means the error model does not hold for this
// search, which is then evaluated at every position (flag 2).  One workgroup per search.
// ------------------------------------------------------------------------------------------
// one 16-byte store of (index, score bits, ready, flagged) into memory the host polls (sushi_hip_batch_set_early_output)
__device__ __forceinline__ void early_store(int4* p, int idx, int score_bits, int ready, int flagged) {
    typedef int i4v __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(i4v{idx, score_bits, ready, flagged}, reinterpret_cast<i4v*>(p));
}
constexpr int RCAP = 128;
constexpr int RAUD = AUDIT_RUNS * FFT_AUDIT;      // audited non-candidate positions per search
constexpr int RENT = RCAP + RAUD;                // list entries: candidates, then audit positions
constexpr int REFINE_THREADS = 256;              // thread <-> (task, chunk): the usual search (its candidate + one audit stretch, a 3 s pattern = 71 chunks) in one round

// Entries [0, n_cand) of the list are candidates; entries [n_cand, n_all) are audit positions: NOT selected, with their plain
// f32 scores, evaluated like the others and only checked against the bound (the two-sided check of the error model).
// Work is handed out as tasks: one candidate, one audit position, or a whole audit run of FFT_AUDIT consecutive positions
// (`tent` = first entry, `tlen` = entries).
template <typename T>
__device__ __forceinline__ void refine_body(const RefineParams& a, const int s_idx, const SearchDesc& sd,
                                            const unsigned long long* list, const int* lpair,
                                            const unsigned long long* rows, const int n_cand, const int n_all,
                                            const short* tent, const short* tlen, const int n_tasks, double* part,
                                            unsigned long long* rkey, float* rerr, int* violated, unsigned* wg_ratio) {
    const int tid = threadIdx.x;
    const int M = sd.tmpl_len;
    const int n_chunks = (M + XM - 1) / XM;
    const T* __restrict__ Tp = (const T*)a.r.src_raw + sd.tmpl_off;
    const T* __restrict__ Wp = (const T*)a.r.dst_raw + sd.win_start;
    const TemplStats ts = templ_stats(a.r.src_s1, a.r.src_s2, sd.tmpl_off, M, a.r.centre);
    const double* __restrict__ w1 = a.r.dst_s1 + sd.win_start;
    const double* __restrict__ w2 = a.r.dst_s2 + sd.win_start;
    const bool ccoeff = a.method == SUSHI_HIP_METHOD_CCOEFF_NORMED;
    auto finish = [&](const int k, const double corr_u) {
        const unsigned p = key_pos(list[k]);
        float score, ranked;                                    // cv2's float32; what the ranking stage approximates
        if (ccoeff) {
            score = finish_ccoeff_normed(corr_u, w1[p + M] - w1[p], w2[p + M] - w2[p], ts, M);
            ranked = 1.0f - score;                              // arg-max as an arg-min (ifft_kernel ranks 1 - score)
            rkey[k] = make_key_max(score, p);
        } else {
            score = score_exact(corr_u, ts, w2, (int64_t)p, M);
            ranked = score;
            rkey[k] = make_key(score, p);
        }
        const unsigned long long eb = rows[(size_t)lpair[k] * FFT_ROW + FFT_CAND + 1];
        const float e_pair = __uint_as_float((unsigned)(eb & 0xffffffffull));
        const float e_model = __uint_as_float((unsigned)(eb >> 32));
        const float lb = key_score(list[k]);
        float err = 0.f;
        if (k >= n_cand) {                                      // a position that was NOT selected: its plain f32 score
            err = fabsf(lb - ranked);
            if (err > e_pair * 1.001f + 1e-7f) *violated = 1;   // the model failed where nobody was looking: every position
            if (e_model > 0.f) atomicMax(&wg_ratio[1], __float_as_uint(err / e_model));
        } else if (lb > 0.f) {                                  // the f32 score itself (a bound clamped at 0 lost it)
            err = fabsf((lb + e_pair) - ranked);
            if (err > e_pair * 1.001f + 1e-7f) *violated = 1;
            if (e_model > 0.f) atomicMax(&wg_ratio[0], __float_as_uint(err / e_model));
        }
        rerr[k] = err;
    };
    // thread <-> (task, chunk): R tasks at a time (the usual search's two -- its candidate and the sixteen audit positions of its
    // one transformed pair -- in one round); then one thread per (task, position) adds its chunk sums in order.  Patterns of
    // more chunks than threads: one task at a time, chunk groups.
    constexpr int nthr = REFINE_THREADS;
    const bool small = n_chunks <= nthr;
    // (a round finishes one task per RAUD threads: a short pattern -- fewer chunks than RAUD -- is limited by that, not by its chunks)
    const int R = small ? min(nthr / n_chunks, nthr / RAUD) : 1;
    const int cpr = small ? n_chunks : nthr;                     // chunks per round and task
    for (int t0 = 0; t0 < n_tasks; t0 += R) {
        const int slot = small ? tid / n_chunks : 0;
        const int t = t0 + slot;
        const bool have = slot < R && t < n_tasks;
        const int64_t p = have ? (int64_t)key_pos(list[tent[t]]) : 0;
        const bool wide = have && tlen[t] > FFT_AUDIT;            // an audit stretch of RAUD positions
        double tot = 0.0;
        for (int c0 = 0; c0 < n_chunks; c0 += cpr) {
            const int cl = small ? tid - slot * n_chunks : tid;   // chunk inside the round
            const int ch = c0 + cl;
            if (have && ch < n_chunks) {
                const int m0 = ch * XM;
                const int64_t room = a.r.dst_len - (sd.win_start + p + m0);
                if (wide) {
                    double v[RAUD];
                    chunk_run<T, RAUD>(Tp + m0, Wp + p + m0, min(XM, M - m0), room, v);
#pragma unroll
                    for (int q = 0; q < RAUD; ++q) part[RAUD * tid + q] = v[q];
                } else {
                    double v[FFT_AUDIT];
                    chunk_run<T, FFT_AUDIT>(Tp + m0, Wp + p + m0, min(XM, M - m0), room, v);
#pragma unroll
                    for (int q = 0; q < FFT_AUDIT; ++q) part[RAUD * tid + q] = v[q];
                }
            }
            __syncthreads();
            if (tid < RAUD * R) {
                const int fs = tid / RAUD, q = tid % RAUD;
                if (t0 + fs < n_tasks && q < tlen[t0 + fs]) {
                    const int cn = min(cpr, n_chunks - c0);
                    for (int c = 0; c < cn; ++c) tot += part[RAUD * (fs * cpr + c) + q];
                }
            }
            __syncthreads();
        }
        if (tid < RAUD * R) {
            const int fs = tid / RAUD, q = tid % RAUD;
            if (t0 + fs < n_tasks && q < tlen[t0 + fs]) finish(tent[t0 + fs] + q, tot);
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(REFINE_THREADS, 3)
void refine_kernel(RefineParams a) {
    __shared__ unsigned long long list[RENT], rkey[RENT];
    __shared__ int lpair[RENT];
    __shared__ float rerr[RENT];
    __shared__ short tent[RENT], tlen[RENT];
    __shared__ double part[RAUD * REFINE_THREADS];
    __shared__ int cnt, ovf, violated, n_all_s, n_tasks_s;
    __shared__ unsigned wg_ratio[2];             // this search's largest error / bound ratios (float bits): candidates, audit
    constexpr int TMASK_PAIRS = 2048;            // which of the search's first 2048 pairs were transformed (the audit picks among them)
    __shared__ unsigned tmask[TMASK_PAIRS / 32];
    const int tid = threadIdx.x;
    const int s_idx = a.first_search + blockIdx.x;
    const SearchDesc sd = a.searches[s_idx];
    const FftLayout lay = fft_layout(sd.win_start, sd.n_pos, sd.tmpl_len);
    // (violated from the start: ifft_kernel found a pair of this search whose lower bound is above one of its real scores --
    // the exclusion cannot be trusted for it, every position is evaluated)
    if (tid == 0) { cnt = 0; ovf = 0; violated = (a.viol && a.viol[s_idx]) ? 1 : 0; wg_ratio[0] = 0u; wg_ratio[1] = 0u; }
    if (tid < TMASK_PAIRS / 32) tmask[tid] = 0u;
    __syncthreads();
    // none: TM_CCOEFF_NORMED with every window uncertain -- then every listed position is a candidate
    const float U = a.gkeys[s_idx] == NO_KEY ? 4.0f : key_score(a.gkeys[s_idx]);
    const unsigned long long* __restrict__ rows = a.cand + (size_t)(sd.first_pair - a.sub_first_pair) * FFT_ROW;
    const float* __restrict__ plb = a.pair_lb + (sd.first_pair - a.sub_first_pair);
    for (int i = tid; i < lay.n_pairs; i += REFINE_THREADS) {
        // (a transformed pair left its error bound: noted here, by every thread for its own pairs, for the audit's choice below --
        // one thread walking the rows one dependent load at a time was a seventh of this kernel)
        if (i < TMASK_PAIRS && rows[(size_t)i * FFT_ROW + FFT_CAND + 1] != NO_KEY) atomicOr(&tmask[i >> 5], 1u << (i & 31));
        if (!(plb[i] <= U)) continue;                         // no position of this pair can be the extremum
        for (int slot = 0; slot <= FFT_CAND; ++slot) {          // (slots behind FFT_CAND: the pair's error bound / audit run)
            const unsigned long long key = rows[(size_t)i * FFT_ROW + slot];
            if (key != NO_KEY && key_score(key) <= U) {
                if (slot == FFT_CAND) {
                    ovf = 1;                                   // that pair had more candidate positions than slots
                } else {
                    const int k = atomicAdd(&cnt, 1);
                    if (k < RCAP) { list[k] = key; lpair[k] = i; } else ovf = 1;
                }
            }
        }
    }
    __syncthreads();
    const int n = cnt < RCAP ? cnt : RCAP;
    if (a.method == SUSHI_HIP_METHOD_CCOEFF_NORMED) {
        // a pattern without variance: cv2 returns a result of all ones, whose first arg-max is position 0
        const TemplStats ts = templ_stats(a.r.src_s1, a.r.src_s2, sd.tmpl_off, sd.tmpl_len, a.r.centre);
        if (ts.flat) {
            if (tid == 0) {
                a.keys[s_idx] = make_key_max(1.0f, 0u); a.gkeys[s_idx] = 0ull;
                if (a.early) early_store(a.early + s_idx, 0, __float_as_int(key_score_max(make_key_max(1.0f, 0u))), 1, 0);
            }
            return;
        }
    }
    if (!ovf) {
        // AUDIT_RUNS x FFT_AUDIT positions per search that are NOT candidates: audit runs of the pairs that were transformed (a pair
        // the bound excluded has no scores), from as many different pairs as there are -- starting at a hash of the search index;
        // evaluated exactly like the candidates and held to the same bound (the check of the error model where it was not already
        // believed).  A run of FFT_AUDIT valid positions is one task; a run cut by the window's edge gives its valid positions
        // one by one.
        for (int t = tid; t < n; t += REFINE_THREADS) { tent[t] = (short)t; tlen[t] = 1; }
        if (tid == 0) {
            int n_all = n, n_tasks = n;
            const unsigned h = (unsigned)s_idx * 2654435761u;
            const int pa0 = (int)((h >> 8) % (unsigned)lay.n_pairs);
            int sel[AUDIT_RUNS], n_sel = 0;
            for (int j = 0; j < lay.n_pairs && n_sel < AUDIT_RUNS; ++j) {
                const int pa = pa0 + j < lay.n_pairs ? pa0 + j : pa0 + j - lay.n_pairs;
                const bool tr = pa < TMASK_PAIRS ? ((tmask[pa >> 5] >> (pa & 31)) & 1u) != 0u : rows[(size_t)pa * FFT_ROW + FFT_CAND + 1] != NO_KEY;
                if (tr) sel[n_sel++] = pa;                                  // transformed: it left its error bound
            }
            bool wide = false;
            if (n_sel >= 1) {
                // the four runs of ONE pair are sixteen consecutive positions: one task, the window loads of ONE position -- a
                // quarter of what four runs from four pairs cost (the pair is picked by the search's hash among the transformed
                // ones: over the searches of a batch every kind of pair is looked at).  A pair at the window's edge may not hold
                // all sixteen: then the runs are taken one by one, from as many pairs as there are.
                const int pick = sel[(h >> 3) % (unsigned)n_sel];
                wide = true;
                for (int q = 0; q < RAUD; ++q) wide = wide && rows[(size_t)pick * FFT_ROW + FFT_CAND + 2 + q] != NO_KEY;
                if (wide) {
                    tent[n_tasks] = (short)n_all; tlen[n_tasks] = RAUD; ++n_tasks;
                    for (int q = 0; q < RAUD; ++q) { list[n_all] = rows[(size_t)pick * FFT_ROW + FFT_CAND + 2 + q]; lpair[n_all] = pick; ++n_all; }
                }
            }
            for (int r = 0; r < AUDIT_RUNS && n_sel > 0 && !wide; ++r) {
                const int pa = sel[r % n_sel], run = r / n_sel;
                unsigned long long key[FFT_AUDIT];
                int valid = 0;
                for (int q = 0; q < FFT_AUDIT; ++q) {
                    key[q] = rows[(size_t)pa * FFT_ROW + FFT_CAND + 2 + run * FFT_AUDIT + q];
                    valid += key[q] != NO_KEY ? 1 : 0;
                }
                if (valid == FFT_AUDIT) { tent[n_tasks] = (short)n_all; tlen[n_tasks] = FFT_AUDIT; ++n_tasks; }
                for (int q = 0; q < FFT_AUDIT; ++q) {
                    if (key[q] == NO_KEY) continue;
                    if (valid != FFT_AUDIT) { tent[n_tasks] = (short)n_all; tlen[n_tasks] = 1; ++n_tasks; }
                    list[n_all] = key[q]; lpair[n_all] = pa; ++n_all;
                }
            }
            n_all_s = n_all; n_tasks_s = n_tasks;
        }
        __syncthreads();
        const int n_all = n_all_s, n_tasks = n_tasks_s;
        if (a.r.dtype == SUSHI_HIP_F32) refine_body<float>(a, s_idx, sd, list, lpair, rows, n, n_all, tent, tlen, n_tasks, part, rkey, rerr, &violated, wg_ratio);
        else refine_body<uint8_t>(a, s_idx, sd, list, lpair, rows, n, n_all, tent, tlen, n_tasks, part, rkey, rerr, &violated, wg_ratio);
        __syncthreads();
        if (tid == 0 && n_all > n) atomicAdd(&a.counters->audited, (unsigned long long)(n_all - n));
    }
    if (ovf || violated) {
        // this search goes to the collection pass: list its pairs that can hold a candidate (every pair when every position is
        // to be evaluated) -- what collect_kernel's workgroups stride over
        for (int i = tid; i < lay.n_pairs; i += REFINE_THREADS)
            if (violated || plb[i] <= U) a.citems[atomicAdd(a.n_citems, 1)] = (sd.first_pair - a.sub_first_pair) + i;
    }
    if (tid == 0) {
        // the run's maxima: one global atomic per search only where it raises the value -- every workgroup hitting the same two
        // words with an atomic each was most of this kernel's time (same-address atomics serialise in the L2)
        if (wg_ratio[0] > *(volatile uint32_t*)&a.counters->max_ratio_bits) atomicMax(&a.counters->max_ratio_bits, wg_ratio[0]);
        if (wg_ratio[1] > *(volatile uint32_t*)&a.counters->max_ratio_audit_bits) atomicMax(&a.counters->max_ratio_audit_bits, wg_ratio[1]);
        if (ovf || violated) {
            // keys[s_idx] stays NO_KEY for exact_tiles_kernel; gkeys[s_idx] keeps the threshold collect_kernel needs
            a.flags[s_idx] = violated ? 2 : 1;
            a.flag_list[atomicAdd(&a.sub->sub_flagged, 1)] = s_idx;
            atomicAdd(&a.counters->n_flagged, 1);
            if (violated) atomicAdd(&a.counters->n_all_positions, 1);
            if (a.early) early_store(a.early + s_idx, 0, 0, 1, 1);      // ready, flagged: the answer comes with the run's last kernel
        } else {
            unsigned long long best = NO_KEY;
            float best_err = 0.f;
            for (int k = 0; k < n; ++k)
                if (rkey[k] < best) { best = rkey[k]; best_err = rerr[k]; }
            a.keys[s_idx] = best;
            // diagnostics: how far the ranking stage was off at the position that won
            a.gkeys[s_idx] = (unsigned long long)__float_as_uint(best_err);
            // the answer as unpack_keys_kernel will write it, NOW, in one 16-byte store (sushi_hip_batch_set_early_output)
            if (a.early) {
                const float sc = a.method == SUSHI_HIP_METHOD_CCOEFF_NORMED ? key_score_max(best) : key_score(best);
                early_store(a.early + s_idx, (int)key_pos(best), __float_as_int(sc), 1, 0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Stream preparation: centred float32 copy + float64 exclusive prefix sums of xc and xc^2.
// Three passes over blocks of PB samples (block totals -> scan of totals -> in-block scan).
// ------------------------------------------------------------------------------------------
constexpr int PB_THREADS = 256;
constexpr int PB_PER_THREAD = 16;
constexpr int PB = PB_THREADS * PB_PER_THREAD;   // 4096 samples per block
static_assert(PB == FFT_HOP, "the relative prefix sums are per FFT block");

template <typename T> __device__ __forceinline__ float centred(T x);
template <> __device__ __forceinline__ float centred<float>(float x) { return x - 0.5f; }
template <> __device__ __forceinline__ float centred<uint8_t>(uint8_t x) { return (float)((int)x - 128); }

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
            const T x = raw[e];
            const double u = (double)x;                                   // the sample as it is
            xc[e] = centred<T>(x);                                        // what the direct kernel multiplies
            s1 += u;
            s2 += u * u;
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

// single workgroup: in-place exclusive scan of the per-block totals of NA arrays; entry [nb] of each
// receives the grand total, so that bs[b] = prefix sum at sample min(b * PB, n) for b = 0 .. nb
template <int NA>
__global__ __launch_bounds__(1024)
void scan_blocksums_kernel(double* __restrict__ bs, int stride, int nb) {
    __shared__ double wt[NA][16];
    __shared__ double carry[NA];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < NA) carry[tid] = 0.0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int k = base + tid;
        double v[NA], e[NA], o[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            v[a] = k < nb ? bs[a * stride + k] : 0.0;
            double t;
            e[a] = wave_excl_scan(v[a], &t);
            if (lane == 0) wt[a][wv] = t;
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            o[a] = carry[a];
            for (int w = 0; w < wv; ++w) o[a] += wt[a][w];
            if (k < nb) bs[a * stride + k] = o[a] + e[a];
        }
        __syncthreads();
        if (tid == 1023) {
#pragma unroll
            for (int a = 0; a < NA; ++a) carry[a] = o[a] + e[a] + v[a];
        }
        __syncthreads();
    }
    if (tid < NA) bs[tid * stride + nb] = carry[tid];
}

// prefix sums s1 = sum x, s2 = sum x^2 of the samples as they are (float64, absolute: exact for uint8) and,
// for the FFT path's scoring, s2 again as float32 relative to the base of the sample's PB-block:
//     s2[e] = base2[e / PB] + urel[e]          (e = 0 .. n)
template <typename T>
__global__ __launch_bounds__(PB_THREADS)
void final_scan_kernel(const T* __restrict__ raw, int64_t n, const double* __restrict__ bs1,
                       const double* __restrict__ bs2, double* __restrict__ s1, double* __restrict__ s2,
                       float* __restrict__ urel, float* __restrict__ usrel) {
    // A thread scans PB_PER_THREAD consecutive samples, but global memory is touched a workgroup-wide row at a
    // time: samples come in and prefix values go out through a padded LDS tile (index + index / 16: the
    // 16-element runs of neighbouring threads start in different banks).
    __shared__ double tile[PB + PB / PB_PER_THREAD];
    __shared__ double w1[PB_THREADS / 64], w2[PB_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t blk = (int64_t)blockIdx.x * PB;
    auto pad = [](const int i) { return i + i / PB_PER_THREAD; };
#pragma unroll
    for (int k = 0; k < PB_PER_THREAD; ++k) {
        const int i = k * PB_THREADS + tid;                              // coalesced
        const int64_t e = blk + i;
        tile[pad(i)] = e < n ? (double)raw[e] : 0.0;
    }
    __syncthreads();
    double v[PB_PER_THREAD];
    double l1 = 0.0, l2 = 0.0;
#pragma unroll
    for (int k = 0; k < PB_PER_THREAD; ++k) {
        v[k] = tile[pad(tid * PB_PER_THREAD + k)];
        l1 += v[k];
        l2 += v[k] * v[k];
    }
    double t1, t2;
    double e1 = wave_excl_scan(l1, &t1);
    double e2 = wave_excl_scan(l2, &t2);
    if (lane == 0) { w1[wv] = t1; w2[wv] = t2; }
    __syncthreads();                                                     // also: everyone has read its samples
    for (int w = 0; w < wv; ++w) { e1 += w1[w]; e2 += w2[w]; }           // prefix inside the block, before the thread's run
    const double o1 = bs1[blockIdx.x], o2 = bs2[blockIdx.x];             // block bases
    if (blockIdx.x == 0 && tid == 0) {
        s1[0] = 0.0; s2[0] = 0.0;
        if (n % PB == 0) { urel[n] = 0.f; usrel[2 * n] = 0.f; usrel[2 * n + 1] = 0.f; }   // sample n opens a block of its own: base[n / PB] = total
    }
    // s1[e + 1], s2[e + 1] (inclusive sums) and urel[e] (exclusive, relative to the block), one array at a time
    {
        double r = e1;
#pragma unroll
        for (int k = 0; k < PB_PER_THREAD; ++k) { r += v[k]; tile[pad(tid * PB_PER_THREAD + k)] = o1 + r; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PB_PER_THREAD; ++k) {
        const int i = k * PB_THREADS + tid;
        if (blk + i < n) s1[blk + i + 1] = tile[pad(i)];
    }
    __syncthreads();
    {
        double r = e2;
#pragma unroll
        for (int k = 0; k < PB_PER_THREAD; ++k) { r += v[k] * v[k]; tile[pad(tid * PB_PER_THREAD + k)] = o2 + r; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PB_PER_THREAD; ++k) {
        const int i = k * PB_THREADS + tid;
        if (blk + i < n) s2[blk + i + 1] = tile[pad(i)];
    }
    __syncthreads();
    // urel[e] (exclusive, relative to the block) and the same for the sum of the samples (TM_CCOEFF_NORMED's window means:
    // s1[e] = base1[e / PB] + srel[e]); the pair goes out twice: urel alone (TM_SQDIFF_NORMED reads nothing else) and
    // interleaved as usrel[e] = (urel[e], srel[e]), so that TM_CCOEFF_NORMED's scoring takes both with ONE 8-byte load per
    // window end instead of two 4-byte ones
    struct f2 { float u, s; };
    f2* __restrict__ ftile = reinterpret_cast<f2*>(tile);
    {
        double r2 = e2, r1 = e1;
#pragma unroll
        for (int k = 0; k < PB_PER_THREAD; ++k) {
            ftile[pad(tid * PB_PER_THREAD + k)] = f2{(float)r2, (float)r1};
            r2 += v[k] * v[k];
            r1 += v[k];
        }
    }
    __syncthreads();
    f2* __restrict__ us = reinterpret_cast<f2*>(usrel);
#pragma unroll
    for (int k = 0; k < PB_PER_THREAD; ++k) {
        const int i = k * PB_THREADS + tid;
        // e == n inside this block (n % PB != 0): samples past the end are zeros, so the running sum there is the total
        if (blk + i <= n) {
            const f2 x = ftile[pad(i)];
            urel[blk + i] = x.u;
            us[blk + i] = x;
        }
    }
}

// s2 and s1 at every COARSE_G-th sample (entries past the end: the totals) -- a table small enough to live in the L2s, from which
// bound_kernel takes a lower bound of the window energies of a whole block pair
__global__ void coarse_prefix_kernel(const double* __restrict__ s1, const double* __restrict__ s2, int64_t n, int64_t nc,
                                     double* __restrict__ coarse) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nc) {
        const int64_t e = j * COARSE_G < n ? j * COARSE_G : n;
        coarse[j] = s2[e];
        coarse[nc + j] = s1[e];
    }
}

// What the FFT path needs to know about a stream as a whole (sushi_fft.hip, "packed halves"): the constant its block
// spectra are centred by -- the stream's own mean, as a float: any constant is exact (sum T I = sum T (I - c) + c sum T),
// the mean keeps DC out of the products whatever level the data sits at -- and the largest centred energy of FFT_STEP + 1
// consecutive blocks (what one block pair's transform can hold at most: the scale of the packed products is derived
// from it).  One workgroup; bs2 / bs1 are the scanned block bases of sum x^2 / sum x.
__global__ __launch_bounds__(1024)
void fft_stats_kernel(const double* __restrict__ bs2, const double* __restrict__ bs1, int nb, int64_t n, double* __restrict__ stats) {
    __shared__ double red[16];
    const int tid = threadIdx.x;
    const double c = (double)(float)(bs1[nb] / (double)n);
    double emax = 0.0;
    for (int j = tid; j < nb; j += 1024) {
        const int je = min(j + FFT_STEP + 1, nb);
        const int64_t lo = (int64_t)j * PB, hi = min((int64_t)je * PB, n);
        const double e = (bs2[je] - bs2[j]) - 2.0 * c * (bs1[je] - bs1[j]) + c * c * (double)(hi - lo);
        emax = e > emax ? e : emax;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const double o = __shfl_down(emax, d, 64); emax = o > emax ? o : emax; }
    if ((tid & 63) == 0) red[tid >> 6] = emax;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) emax = red[w] > emax ? red[w] : emax;
        stats[0] = emax;
        stats[1] = c;
    }
}

inline int launch_ok() { return hipGetLastError() == hipSuccess ? SUSHI_HIP_OK : SUSHI_HIP_ELAUNCH; }

struct Variant { int waves, nb; };
constexpr Variant kVariants[] = {{1, 1}, {4, 1}, {4, 4}};
constexpr int kNumVariants = 3;

}  // namespace

namespace sushi {

int direct_variant_count() { return kNumVariants; }

int direct_variant_tile(int variant) {
    if (variant < 0 || variant >= kNumVariants) return 0;
    return kVariants[variant].waves * kVariants[variant].nb * 1024;
}

int launch_unpack(const unsigned long long* keys_dev, int n, int method, int32_t* out_idx_dev, float* out_score_dev,
                  int32_t* out_packed_dev, hipStream_t st) {
    hipLaunchKernelGGL(unpack_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, st, keys_dev, n, method, out_idx_dev,
                       out_score_dev, reinterpret_cast<int2*>(out_packed_dev));
    return launch_ok();
}

int launch_direct(const StreamRefs& r, const SearchDesc* searches_dev, int n_search, int n_tiles, int variant, int method,
                  unsigned long long* keys_dev, int32_t* out_idx_dev, float* out_score_dev, int32_t* out_packed_dev, hipStream_t st) {
    if (n_tiles < n_search || variant < 0 || variant >= kNumVariants) return SUSHI_HIP_EINVAL;
    if (method != SUSHI_HIP_METHOD_SQDIFF_NORMED && method != SUSHI_HIP_METHOD_CCOEFF_NORMED) return SUSHI_HIP_EINVAL;
    if (hipMemsetAsync(keys_dev, 0xff, (size_t)n_search * sizeof(uint64_t), st) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    MatchArgs a;
    a.dst_xc = r.dst_xc; a.dst_s1 = r.dst_s1; a.dst_s2 = r.dst_s2; a.dst_len = r.dst_len;
    a.src_xc = r.src_xc; a.src_s1 = r.src_s1; a.src_s2 = r.src_s2; a.src_len = r.src_len;
    a.centre = r.centre; a.searches = searches_dev; a.n_search = n_search; a.n_tiles = n_tiles;
    a.keys = keys_dev; a.method = method;
    switch (variant) {
        case 0: hipLaunchKernelGGL((match_sqdiff_f32_kernel<1, 1>), dim3(n_tiles), dim3(64), 0, st, a); break;
        case 1: hipLaunchKernelGGL((match_sqdiff_f32_kernel<4, 1>), dim3(n_tiles), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((match_sqdiff_f32_kernel<4, 4>), dim3(n_tiles), dim3(256), 0, st, a); break;
    }
    if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
    return launch_unpack(keys_dev, n_search, method, out_idx_dev, out_score_dev, out_packed_dev, st);
}

__global__ __launch_bounds__(256)
void fill_ranges_kernel(FillArgs a) {
    const unsigned stride = gridDim.x * 256u;
    for (int r = 0; r < a.n; ++r) {
        uint32_t* __restrict__ p = a.p[r];
        const uint32_t v = a.value[r];
        // (16-byte stores where the range allows: every range starts 256-byte aligned)
        const unsigned quads = a.words[r] >> 2;
        uint4* __restrict__ p4 = reinterpret_cast<uint4*>(p);
        for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < quads; i += stride) p4[i] = uint4{v, v, v, v};
        for (unsigned i = (quads << 2) + blockIdx.x * 256u + threadIdx.x; i < a.words[r]; i += stride) p[i] = v;
    }
}

int launch_fill(const FillArgs& a, hipStream_t st) {
    uint64_t most = 0;
    for (int r = 0; r < a.n; ++r) most = a.words[r] > most ? a.words[r] : most;
    const unsigned grid = (unsigned)std::min<uint64_t>(1024, std::max<uint64_t>(1, (most / 4 + 255) / 256));
    hipLaunchKernelGGL(fill_ranges_kernel, dim3(grid), dim3(256), 0, st, a);
    return launch_ok();
}

int launch_refine(const RefineParams& p, hipStream_t st) {
    hipLaunchKernelGGL(refine_kernel, dim3(p.n_sub), dim3(REFINE_THREADS), 0, st, p);
    return launch_ok();
}

int launch_tiles(const TileParams& p, hipStream_t st) {
    if (p.r.dtype == SUSHI_HIP_F32) hipLaunchKernelGGL(exact_tiles_kernel<float>, dim3(2048), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(exact_tiles_kernel<uint8_t>, dim3(2048), dim3(256), 0, st, p);
    return launch_ok();
}

}  // namespace sushi

using namespace sushi;

namespace {

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// where the parts of a prepared stream go inside the caller's buffer
struct StreamLayout { size_t xc, s1, s2, urel, srel, base, base_bytes, coarse, spec, total; };

StreamLayout stream_layout(int64_t n, int searchable) {
    StreamLayout l;
    const int64_t nb = (n + PB - 1) / PB;
    size_t o = 0;
    l.xc = o; o += align_up((size_t)n * sizeof(float), 256);
    l.s1 = o; o += align_up((size_t)(n + 1) * sizeof(double), 256);
    l.s2 = o; o += align_up((size_t)(n + 1) * sizeof(double), 256);
    l.urel = o; o += align_up((size_t)(n + 1) * sizeof(float), 256);
    l.srel = o; o += align_up((size_t)(n + 1) * 2 * sizeof(float), 256);     // usrel: (urel, srel) interleaved
    l.base_bytes = (size_t)(2 * (nb + 1) + 2) * sizeof(double);  // block bases of sum x^2, then of sum x, then the FFT path's stats
    l.base = o; o += align_up(l.base_bytes, 256);
    l.coarse = o; o += align_up((size_t)2 * (size_t)(n / COARSE_G + 2) * sizeof(double), 256);
    l.spec = o; o += searchable ? align_up(sushi_hip_stream_spectra_bytes(n), 256) : 0;
    l.total = o;
    return l;
}

}  // namespace

extern "C" {

int sushi_hip_abi_version(void) { return SUSHI_HIP_ABI_VERSION; }

const char* sushi_hip_strerror(int code) {
    switch (code) {
        case SUSHI_HIP_OK: return "ok";
        case SUSHI_HIP_EINVAL: return "invalid argument";
        case SUSHI_HIP_EALIGN: return "device pointer not aligned";
        case SUSHI_HIP_ELAUNCH: return "HIP launch failed";
        case SUSHI_HIP_ENOSPACE: return "buffer or workspace too small";
        case SUSHI_HIP_ENODEV: return "no gfx950 device";
        case SUSHI_HIP_ENOMEM: return "out of host memory";
        case SUSHI_HIP_EINTERNAL: return "internal error (a C++ exception was caught at the boundary)";
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

double sushi_hip_centre(int dtype) { return dtype == SUSHI_HIP_U8 ? 128.0 : 0.5; }

size_t sushi_hip_stream_bytes(int64_t n, int dtype, int searchable) {
    if (n <= 0 || (dtype != SUSHI_HIP_U8 && dtype != SUSHI_HIP_F32)) return 0;
    return stream_layout(n, searchable).total;
}

int sushi_hip_stream_create(const void* raw_dev, int dtype, int64_t n, int searchable, void* mem_dev, size_t mem_bytes,
                            void* hip_stream, SushiHipStream** out) {
    if (!raw_dev || !mem_dev || !out || n <= 0) return SUSHI_HIP_EINVAL;
    if (dtype != SUSHI_HIP_U8 && dtype != SUSHI_HIP_F32) return SUSHI_HIP_EINVAL;
    if (((uintptr_t)mem_dev & 255) || (dtype == SUSHI_HIP_F32 && ((uintptr_t)raw_dev & 3))) return SUSHI_HIP_EALIGN;
    const StreamLayout l = stream_layout(n, searchable);
    if (mem_bytes < l.total) return SUSHI_HIP_ENOSPACE;
    const int64_t nb64 = (n + PB - 1) / PB;
    if (nb64 > 0x7ffffffe) return SUSHI_HIP_EINVAL;
    const int nb = (int)nb64;
    SushiHipStream* s = new (std::nothrow) SushiHipStream();
    if (!s) return SUSHI_HIP_EINVAL;
    char* m = (char*)mem_dev;
    s->raw = raw_dev; s->dtype = dtype; s->n = n;
    s->xc = (float*)(m + l.xc); s->s1 = (double*)(m + l.s1); s->s2 = (double*)(m + l.s2);
    s->urel = (float*)(m + l.urel); s->usrel = (float*)(m + l.srel); s->base = (double*)(m + l.base); s->base_bytes = l.base_bytes;
    s->spec = nullptr; s->spec_low = nullptr; s->znorm_rest = nullptr; s->norm_stride = 0; s->spec_bytes = 0; s->blocks = nb; s->stats = s->base + 2 * (nb + 1);
    s->coarse = (double*)(m + l.coarse); s->nc = n / COARSE_G + 2;
    hipStream_t st = (hipStream_t)hip_stream;
    double* bs2 = s->base;                       // block bases of sum x^2 (what the FFT path's scoring reads)
    double* bs1 = s->base + (nb + 1);            // block bases of sum x
    if (dtype == SUSHI_HIP_F32)
        hipLaunchKernelGGL(centre_blocksum_kernel<float>, dim3(nb), dim3(PB_THREADS), 0, st,
                           (const float*)raw_dev, n, s->xc, bs1, bs2);
    else
        hipLaunchKernelGGL(centre_blocksum_kernel<uint8_t>, dim3(nb), dim3(PB_THREADS), 0, st,
                           (const uint8_t*)raw_dev, n, s->xc, bs1, bs2);
    int rc = launch_ok();
    if (rc == SUSHI_HIP_OK) {
        hipLaunchKernelGGL(scan_blocksums_kernel<2>, dim3(1), dim3(1024), 0, st, s->base, nb + 1, nb);
        rc = launch_ok();
    }
    if (rc == SUSHI_HIP_OK) {
        if (dtype == SUSHI_HIP_F32)
            hipLaunchKernelGGL(final_scan_kernel<float>, dim3(nb), dim3(PB_THREADS), 0, st, (const float*)raw_dev, n,
                               (const double*)bs1, (const double*)bs2, s->s1, s->s2, s->urel, s->usrel);
        else
            hipLaunchKernelGGL(final_scan_kernel<uint8_t>, dim3(nb), dim3(PB_THREADS), 0, st, (const uint8_t*)raw_dev, n,
                               (const double*)bs1, (const double*)bs2, s->s1, s->s2, s->urel, s->usrel);
        rc = launch_ok();
    }
    if (rc == SUSHI_HIP_OK) {
        hipLaunchKernelGGL(fft_stats_kernel, dim3(1), dim3(1024), 0, st, (const double*)bs2, (const double*)bs1, nb, n, s->stats);
        rc = launch_ok();
    }
    if (rc == SUSHI_HIP_OK) {
        hipLaunchKernelGGL(coarse_prefix_kernel, dim3((unsigned)((s->nc + 255) / 256)), dim3(256), 0, st, (const double*)s->s1,
                           (const double*)s->s2, n, s->nc, s->coarse);
        rc = launch_ok();
    }
    if (rc == SUSHI_HIP_OK && searchable)
        rc = sushi_hip_stream_add_spectra(s, m + l.spec, mem_bytes - l.spec, hip_stream);
    if (rc != SUSHI_HIP_OK) { delete s; return rc; }
    *out = s;
    return SUSHI_HIP_OK;
}

int sushi_hip_stream_view(const SushiHipStream* s, int which, const void** ptr_dev, size_t* bytes) {
    if (!s || !ptr_dev || !bytes) return SUSHI_HIP_EINVAL;
    switch (which) {
        case SUSHI_HIP_VIEW_XC: *ptr_dev = s->xc; *bytes = (size_t)s->n * sizeof(float); break;
        case SUSHI_HIP_VIEW_S1: *ptr_dev = s->s1; *bytes = (size_t)(s->n + 1) * sizeof(double); break;
        case SUSHI_HIP_VIEW_S2: *ptr_dev = s->s2; *bytes = (size_t)(s->n + 1) * sizeof(double); break;
        case SUSHI_HIP_VIEW_UREL: *ptr_dev = s->urel; *bytes = (size_t)(s->n + 1) * sizeof(float); break;
        case SUSHI_HIP_VIEW_BASE: *ptr_dev = s->base; *bytes = (size_t)(s->blocks + 1) * sizeof(double); break;
        case SUSHI_HIP_VIEW_SPECTRA: *ptr_dev = s->spec; *bytes = s->spec_bytes; break;
        case SUSHI_HIP_VIEW_SPECTRA_LOW: *ptr_dev = s->spec_low; *bytes = s->spec ? s->spec_bytes / 4 : 0; break;
        case SUSHI_HIP_VIEW_ZNORM_REST: *ptr_dev = s->znorm_rest; *bytes = s->spec ? (size_t)3 * (size_t)s->norm_stride * sizeof(float) : 0; break;
        case SUSHI_HIP_VIEW_USREL: *ptr_dev = s->usrel; *bytes = (size_t)(s->n + 1) * 2 * sizeof(float); break;
        case SUSHI_HIP_VIEW_BASE1: *ptr_dev = s->base + (s->blocks + 1); *bytes = (size_t)(s->blocks + 1) * sizeof(double); break;
        case SUSHI_HIP_VIEW_COARSE: *ptr_dev = s->coarse; *bytes = (size_t)2 * (size_t)s->nc * sizeof(double); break;
        default: return SUSHI_HIP_EINVAL;
    }
    return SUSHI_HIP_OK;
}

void sushi_hip_stream_destroy(SushiHipStream* s) { delete s; }

}  // extern "C"
