// sushi_amd/csrc/fft_core.hpp -- workgroup FFT (complex f32) for gfx950: N = 2^LOGN points by N/16 threads x 16 points.
//
// Used by the overlap-save form of the template match (DESIGN.md "FFT path"): forward transforms of
// destination-stream blocks and pattern segments, inverse transforms of the per-block products.
// OpenCV's crossCorr() (templmatch.cpp, behind cv2.matchTemplate at reference wav.py:185) is the same
// block-DFT scheme on the CPU.
//
//   Plan<13>:  8192 points,  512 threads, radix plan 8 x 8 x  8 x 16
//   Plan<14>: 16384 points, 1024 threads, radix plan 8 x 8 x 16 x 16      (the product's size)
//
// Stockham autosort.  A pass of radix R with NS = product of the earlier radices processes butterflies
// j = 0 .. N/R-1:
//     inputs   x[j + t*N/R] * w^(t*k),   k = j mod NS,  w = exp(DIR*2*pi*i / (NS*R)),  t = 0..R-1
//     outputs  y[(j - k)*R + k + t*NS]   = R-point DFT of the inputs
// Pass 1 (NS = 1, no twiddles) takes its inputs from registers, and a thread owns the two ADJACENT
// butterflies j = 2*tid, 2*tid + 1: its inputs x[2*tid + b + (N/8)*t] are 16-byte pairs, so that the
// caller can fill them with dwordx4 loads (8-byte global loads reach only ~0.6 of the HBM rate on
// gfx950).  In the other passes a thread owns j = tid (+ NT).  The last pass (radix 16 = the points a
// thread holds) leaves its outputs in registers: thread `tid` ends with X[tid + NT*r], r = 0..15.
//
// Between passes the real parts, then the imaginary parts, go through ONE float buffer of N + N/8 floats
// (36 KB / 72 KB: four / two workgroups per CU).  After a part's loads every thread's registers hold the new
// part next to the OTHER part of the old element set, which is then stored in turn.  Each exchange lays the
// buffer out its own way (its stores and loads agree), chosen so that both the transposing stores and the
// unit-stride loads hit 32 different banks per 32-lane group:
//   after pass 1 : pos(e) = e + (e >> 5)        stores 16 tid + u -> 16 tid + (tid >> 1) + u   (lane stride 16.5)
//   after pass 2 : pos(e) = e + 8 (e >> 6)      stores 64 a + b + 8 t (j = 8 a + b) -> 72 a + b + 8 t
//   after pass 3 : pos(e) = e                   stores 64 R3 w + l + 64 t: lanes are unit stride as they are
// Offsets stay immediates.
//
// Everything here is written against explicit (tid, lds) arguments so that tests/host_fft_check.cpp
// can run the same code on the CPU, one "thread" at a time, with the barriers replaced by loops.
#ifndef SUSHI_FFT_CORE_HPP
#define SUSHI_FFT_CORE_HPP

#ifdef __HIPCC__
#define SUSHI_HD __device__ __forceinline__
#define SUSHI_HHD __host__ __device__ inline
#else
#define SUSHI_HD inline
#define SUSHI_HHD inline
#endif

namespace sushi_fft {

constexpr int PER = 16;          // points per thread, every plan
constexpr int TWIDDLE_N = 16384; // length of the twiddle table exp(-2*pi*i*n/TWIDDLE_N) every plan indexes

template <int LOGN> struct Plan;
template <> struct Plan<13> {
    static constexpr int N = 8192, NT = 512, R3 = 8;
};
template <> struct Plan<14> {
    static constexpr int N = 16384, NT = 1024, R3 = 16;
};
// radices 1, 2 and 4 are the same in both plans
constexpr int R1 = 8, R2 = 8, R4 = 16;

template <int LOGN> constexpr int lds_floats() { return Plan<LOGN>::N + Plan<LOGN>::N / 8; }

struct cpx { float x, y; };

SUSHI_HD cpx cadd(cpx a, cpx b) { return cpx{a.x + b.x, a.y + b.y}; }
SUSHI_HD cpx csub(cpx a, cpx b) { return cpx{a.x - b.x, a.y - b.y}; }
SUSHI_HD cpx cmul(cpx a, cpx b) { return cpx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
SUSHI_HD cpx cconj(cpx a) { return cpx{a.x, -a.y}; }
SUSHI_HD float fmaf_(float a, float b, float c) {
#ifdef __HIPCC__
    return __builtin_fmaf(a, b, c);
#else
    return a * b + c;
#endif
}

// lo = e + w o, hi = e - w o for w = exp(DIR * 2*pi*i * q/32), q = 0..15 a constant after unrolling: the twiddle is folded into
// the butterfly's own multiply-adds -- six operations for a general root (hi = 2 e - lo) and for the odd multiples of pi/4,
// four for 1 and +-i, where a multiplication followed by an addition and a subtraction takes eight
template <int DIR>
SUSHI_HD void bfly_root32(const cpx e, const cpx o, int q, cpx& lo, cpx& hi) {
    const float H = 0.70710678118654752440f;
    const float C[9] = {1.0f, 0.98078528040323044913f, 0.92387953251128673848f, 0.83146961230254523708f, H,
                        0.55557023301960222474f, 0.38268343236508978178f, 0.19509032201612826785f, 0.0f};
    const float s = (float)DIR;
    switch (q) {
        case 0: lo = cadd(e, o); hi = csub(e, o); return;
        case 8: lo = cpx{e.x - s * o.y, e.y + s * o.x}; hi = cpx{e.x + s * o.y, e.y - s * o.x}; return;
        case 4: {
            const float p = o.x - s * o.y, r = o.y + s * o.x;
            lo = cpx{fmaf_(H, p, e.x), fmaf_(H, r, e.y)}; hi = cpx{fmaf_(-H, p, e.x), fmaf_(-H, r, e.y)};
            return;
        }
        case 12: {
            const float p = o.x + s * o.y, r = s * o.x - o.y;
            lo = cpx{fmaf_(-H, p, e.x), fmaf_(H, r, e.y)}; hi = cpx{fmaf_(H, p, e.x), fmaf_(-H, r, e.y)};
            return;
        }
        default: {
            // cos(2 pi q / 32) = C[q] (q <= 8), -C[16 - q] (q > 8); sin = C[8 - q] (q <= 8), C[q - 8] (q > 8)
            const float c = q <= 8 ? C[q] : -C[16 - q], sn = s * (q <= 8 ? C[8 - q] : C[q - 8]);
            lo = cpx{fmaf_(-sn, o.y, fmaf_(c, o.x, e.x)), fmaf_(sn, o.x, fmaf_(c, o.y, e.y))};
            hi = cpx{fmaf_(2.0f, e.x, -lo.x), fmaf_(2.0f, e.y, -lo.y)};
        }
    }
}

// in-place R-point DFT, natural order in and out: v[t] <- sum_u v[u] * exp(DIR*2*pi*i*t*u/R)
template <int R, int DIR>
struct Dft {
    static SUSHI_HD void run(cpx* v) {
        cpx e[R / 2], o[R / 2];
#pragma unroll
        for (int u = 0; u < R / 2; ++u) { e[u] = v[2 * u]; o[u] = v[2 * u + 1]; }
        Dft<R / 2, DIR>::run(e);
        Dft<R / 2, DIR>::run(o);
#pragma unroll
        for (int t = 0; t < R / 2; ++t) bfly_root32<DIR>(e[t], o[t], t * (32 / R), v[t], v[t + R / 2]);
    }
};
template <int DIR>
struct Dft<1, DIR> {
    static SUSHI_HD void run(cpx*) {}
};

// Base twiddles of the twiddled passes for this thread, w^1 = exp(DIR*2*pi*i*k/(NS*R)) with k = j mod NS:
// the butterflies of a thread (j = tid + b*NT) share k in the middle passes, and the last pass has one
// butterfly per thread.  tw[n] = exp(-2*pi*i*n/TWIDDLE_N) (the forward table; the inverse conjugates it).
// Loaded once, before the first barrier, so that no table load sits between two passes.
struct Twiddles { cpx p2, p3, p4; };

template <int LOGN, int DIR>
SUSHI_HD Twiddles load_twiddles(int tid, const cpx* __restrict__ tw) {
    typedef Plan<LOGN> P;
    Twiddles t;
    t.p2 = tw[(tid & 7) * (TWIDDLE_N / (R1 * R2))];             // pass 2: NS = 8
    t.p3 = tw[(tid & 63) * (TWIDDLE_N / (R1 * R2 * P::R3))];    // pass 3: NS = 64
    t.p4 = tw[tid * (TWIDDLE_N / P::N)];                        // pass 4: NS = NT, one butterfly per thread
    if (DIR > 0) { t.p2 = cconj(t.p2); t.p3 = cconj(t.p3); t.p4 = cconj(t.p4); }
    return t;
}

// One pass, register side: twiddle (unless NS == 1) and butterfly the 16 points of this thread.
// v[b*R + t] holds input t of the thread's butterfly b; w1 is the base twiddle (shared by all of them).
// The powers w^t are built one multiplication at a time (w^t = w^(t-1) * w): two live twiddle registers instead of R,
// R - 2 complex multiplications instead of the ~2R of a squaring scheme, and the same rounding error at t = R - 1
// (t errors of the base twiddle plus t - 1 product roundings either way).
template <int R, int NS, int DIR>
SUSHI_HD void pass_compute(cpx* v, const cpx w1) {
    constexpr int NB = PER / R;
    if (NS > 1) {
        cpx wt = w1;
#pragma unroll
        for (int t = 1; t < R; ++t) {
#pragma unroll
            for (int b = 0; b < NB; ++b) v[b * R + t] = cmul(v[b * R + t], wt);
            if (t + 1 < R) wt = cmul(wt, w1);
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) Dft<R, DIR>::run(v + b * R);
}

// Index maps of the whole transform (what the caller needs to load / interpret registers):
//   input : v[b*8 + t] = x[2*tid + b + (N/8)*t]      b = 0..1, t = 0..7   (adjacent pairs: 16-byte loads)
//   output: v[r]       = X[tid + NT*r]               r = 0..15
template <int LOGN>
SUSHI_HD int in_index(int tid, int r) { return 2 * tid + (r / R1) + (Plan<LOGN>::N / R1) * (r % R1); }
template <int LOGN>
SUSHI_HD int out_index(int tid, int r) { return tid + Plan<LOGN>::NT * r; }

template <int IM> SUSHI_HD float part_of(const cpx& c) { return IM ? c.y : c.x; }
template <int IM> SUSHI_HD void set_part(cpx& c, const float e) { if (IM) c.y = e; else c.x = e; }

// outputs of pass EX (1, 2, 3) -> the float buffer, in that exchange's layout
template <int LOGN, int EX, int IM>
SUSHI_HD void split_store(const cpx* v, int tid, float* lds) {
    typedef Plan<LOGN> P;
    if (EX == 1) {                                   // the thread's 16 contiguous elements 16 tid + u
        float* out = lds + 16 * tid + (tid >> 1);
#pragma unroll
        for (int u = 0; u < 16; ++u) out[u] = part_of<IM>(v[u]);
    } else if (EX == 2) {                            // R = 8, NS = 8: base = 64 (j >> 3) + (j & 7), elements base + 8 t
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int j = tid + P::NT * b;
            float* out = lds + 72 * (j >> 3) + (j & 7);
#pragma unroll
            for (int t = 0; t < 8; ++t) out[8 * t] = part_of<IM>(v[b * 8 + t]);
        }
    } else {                                         // R = R3, NS = 64: base = R3 (j - k) + k, k = j & 63, elements base + 64 t
        constexpr int NB = PER / P::R3;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int j = tid + P::NT * b;
            float* out = lds + P::R3 * (j & ~63) + (j & 63);
#pragma unroll
            for (int t = 0; t < P::R3; ++t) out[64 * t] = part_of<IM>(v[b * P::R3 + t]);
        }
    }
}

// inputs of the pass after exchange EX (radix R): x[j + t N/R], j = tid (+ NT)
template <int LOGN, int EX, int IM>
SUSHI_HD void split_load(cpx* v, int tid, const float* lds) {
    typedef Plan<LOGN> P;
    constexpr int R = EX == 1 ? R2 : (EX == 2 ? P::R3 : R4);
    constexpr int NB = PER / R;
    constexpr int S = P::N / R;
    constexpr int STEP = EX == 1 ? S + (S >> 5) : (EX == 2 ? S + 8 * (S >> 6) : S);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = tid + P::NT * b;
        const float* in = lds + (EX == 1 ? j + (j >> 5) : (EX == 2 ? j + 8 * (j >> 6) : j));
#pragma unroll
        for (int t = 0; t < R; ++t) set_part<IM>(v[b * R + t], in[t * STEP]);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Wave plan for 16384 points (the INVERSE transforms of ifft_kernel / collect_kernel): 16 x 16 x 4 x 16, where the
// first three passes stay inside a wave and only the last exchange crosses the workgroup.
//
// Why: LDS traffic and VALU work of a CU hardly overlap on gfx950 (tools/ubench/valu_lds_overlap.hip: 16 dwords per lane
// out and back cost 87 cycles per wave as b32 operations, 59 as b64, and a wave's FMAs next to them only partly hide);
// the 8 x 8 x 16 x 16 plan above moves every point through the LDS three times as single dwords and takes 11 barriers.
// This plan moves every point through the LDS twice, reads it back as 8-byte pairs, and takes 4 barriers:
//
//   index split   n = n1 + 16 n2 (n1 = wave),  n2 = 64 d1 + 4 d2 + d3   (d1: registers, d2 = lane & 15, d3 = lane >> 4)
//   pass 1        16-point DFTs over d1 -> e1                               in registers
//   row exchange  (lane & 15, register) transposed inside every row of 16 lanes, through the wave's OWN 1152 floats of
//                 the buffer (no barrier: a wave's LDS operations execute in order); rows padded to 18 floats so that the
//                 b32 stores are 2-way (free) and the b64 loads conflict-free
//   pass 2        twiddle w256^(e1 d2), 16-point DFTs over d2 -> e2         in registers
//   swap          register bits 3:2 <-> lane bits 5:4 with v_permlane32_swap / v_permlane16_swap (no LDS)
//   pass 3        twiddle w1024^(d3 (e1 + 16 e2)), 4-point DFTs over d3 -> e3
//                 the wave now holds A_n1[k2] = sum_n2 x[n1 + 16 n2] w1024^(n2 k2),  k2 = e1 + 16 e2 + 256 e3
//   wg exchange   to thread k2 = tid, register n1: position 18 k2 + n1 (2-way stores, conflict-free b64 loads)
//   pass 4        twiddle w16384^(n1 k2), 16-point DFTs over n1 -> k1:  thread tid ends with X[tid + 1024 k1]
//
// The input order is whatever makes the first loads coalesced: thread (wave w, lane l) register d1 holds
// x[w + 1024 d1 + 64 (l & 15) + 16 (l >> 4)], and the spectra this transform consumes are STORED in that order
// (`wslot`: mac_kernel only needs rows, pattern spectra and products to agree on one order of the bins).
// ------------------------------------------------------------------------------------------------------------
constexpr int WN = 16384, WNT = 1024;
constexpr int WROW = 18;                       // floats per padded run of 16
constexpr int W_LDS_FLOATS = WROW * 1024;      // 72 KB: the workgroup exchange; a wave's row exchange uses floats [1152 w, 1152 w + 1152)

// index of the 4-bin entry, inside a stored spectrum of N complex values, that holds registers 4u .. 4u+3 of thread `tid`
// (spectra are stored as packed halves: an entry is 16 bytes, one load brings four registers, a wave's load instruction
// one contiguous KiB)
SUSHI_HD int wslot_uint4(int tid, int u) { return (((tid >> 6) * 4 + u) << 6) + (tid & 63); }
// the frequency bin a thread's register d1 holds when it loads a stored spectrum
SUSHI_HD int wbin(int tid, int d1) { return (tid >> 6) + 1024 * d1 + 64 * (tid & 15) + 16 * ((tid >> 4) & 3); }
// complex index, inside a stored spectrum, of bin f
SUSHI_HHD int wslot_of_bin(int f) {
    const int w = f & 15, d3 = (f >> 4) & 3, d2 = (f >> 6) & 15, d1 = f >> 10;
    return (((((w * 4 + (d1 >> 2)) << 6) + 16 * d3 + d2) << 1) + ((d1 >> 1) & 1)) * 2 + (d1 & 1);
}

// ------------------------------------------------------------------------------------------------------------
// First pass of the wave plan on the matrix pipe (inverse transforms whose input arrives as packed halves).
//
// Pass 1 is 64 independent 16-point DFTs per wave (over d1, one per column (d2, d3)): as a matrix product with the data as the
// A operand of v_mfma_f32_16x16x32_f16,
//     D[m][n] = sum_k A[m][k] B[k][n],   m = a column of the group, k = (part, d1): 16 real parts then 16 imaginary parts, n = e1,
// B holding the DFT matrix ([Wr; -Wi] for the real parts of the result, [Wi; Wr] for the imaginary ones; each as the sum of a high
// and a low half, so that the products are exact to float32), the result lands where the wave plan wants it AFTER its row
// exchange: lane (d3 = lane >> 4, e1 = lane & 15) holds d2 = 0 .. 15 in registers -- the exchange through the LDS, pass 1's ~170
// VALU instructions and the unpacking of the halves all go.
//   A operand of lane l = (q = l >> 4, m = l & 15), group g: the 8 halves k = 8 q .. 8 q + 7 of column (g, m), i.e.
//       part = q >> 1 (0: real, 1: imaginary), d1 = 8 (q & 1) + j, j = 0 .. 7, of column d2 = 4 g + (m & 3), d3 = m >> 2.
//   What the lane LOADS is four whole complex bins (a 16-byte entry, as mac_kernel stores them): lanes below 32 the bins
//       d1 = 8 (q & 1) + 0 .. 3 of their column, lanes l + 32 the bins d1 = 8 (q & 1) + 4 .. 7 of the same column; two v_perm_b32 pairs
//       and two v_permlane32_swap per entry turn (re, im) x 4 into (re x 8 | im x 8).
//   D of lane (q', e1), group g, register i: column m = 4 q' + i, i.e. d3 = q', d2 = 4 g + i.
// Stored spectra are then in THIS load order (mslot_of_bin).
// ------------------------------------------------------------------------------------------------------------
// complex index, inside a stored spectrum, of bin f (MFMA load order)
SUSHI_HHD int mslot_of_bin(int f) {
    const int n1 = f & 15, n2 = f >> 4;
    const int d3 = n2 & 3, d2 = (n2 >> 2) & 15, d1 = n2 >> 6;
    const int g = d2 >> 2, m = (d3 << 2) | (d2 & 3);
    const int dq = d1 >> 2, t = d1 & 3;                          // the lane pair (l, l + 32) splits a d1 octet in two quads
    const int q = (dq >> 1) | ((dq & 1) << 1);                   // octet = dq >> 1 = q & 1, upper quad <=> q >> 1
    return ((((n1 * 4 + g) << 6) + 16 * q + m) << 2) + t;
}
// the bin that sub-position t of entry g of thread tid holds when it loads a stored spectrum
SUSHI_HHD int mbin(int tid, int g, int t) {
    const int n1 = tid >> 6, l = tid & 63, q = l >> 4, m = l & 15;
    const int d1 = 8 * (q & 1) + 4 * (q >> 1) + t, d2 = 4 * g + (m & 3), d3 = m >> 2;
    return n1 + 16 * (64 * d1 + 4 * d2 + d3);
}
// B operands of the four products of a group (real parts: high, low; imaginary parts: high, low) for lane l: element j is
// row k = 8 (l >> 4) + j of column n = l & 15.  Filled by dft16_operand(); the device keeps them in a table.
SUSHI_HHD double dft16_operand(int form, int l, int j, int dir) {
    const int k = 8 * (l >> 4) + j, n = l & 15, kk = k & 15, part = k >> 4;
    const double PI = 3.14159265358979323846;
    const double wr = __builtin_cos(2.0 * PI * (double)((n * kk) & 15) / 16.0), wi = (double)dir * __builtin_sin(2.0 * PI * (double)((n * kk) & 15) / 16.0);
    return form == 0 ? (part == 0 ? wr : -wi) : (part == 0 ? wi : wr);       // form 0: real parts of the result, 1: imaginary parts
}

struct WTwiddles { cpx g2, q3, p4; };

template <int DIR>
SUSHI_HD WTwiddles load_wtwiddles(int tid, const cpx* __restrict__ tw) {
    WTwiddles t;
    const int lane = tid & 63;
    t.g2 = tw[(lane & 15) * (TWIDDLE_N / 256)];                          // pass 2: w256^(e1), e1 = lane & 15 after the row exchange
    t.q3 = tw[((lane & 15) + 64 * (lane >> 4)) * (TWIDDLE_N / 1024)];    // pass 3: w1024^(e1 + 64 e2hi), e2hi = lane >> 4 after the swap
    t.p4 = tw[tid * (TWIDDLE_N / WN)];                                   // pass 4: w16384^(k2), k2 = tid
    if (DIR > 0) { t.g2 = cconj(t.g2); t.q3 = cconj(t.q3); t.p4 = cconj(t.p4); }
    return t;
}

// row exchange: element (row, c = lane & 15, register q) -> (row, c' = q, register c)
template <int IM>
SUSHI_HD void w_row_store(const cpx* v, int tid, float* lds) {
    float* out = lds + (tid >> 6) * (64 * WROW) + ((tid >> 4) & 3) * (16 * WROW) + (tid & 15);
#pragma unroll
    for (int q = 0; q < 16; ++q) out[WROW * q] = part_of<IM>(v[q]);
}
template <int IM>
SUSHI_HD void w_row_load(cpx* v, int tid, const float* lds) {
    const float* in = lds + (tid >> 6) * (64 * WROW) + ((tid >> 4) & 3) * (16 * WROW) + WROW * (tid & 15);
#pragma unroll
    for (int i = 0; i < 8; ++i) {                                         // 8-byte aligned pairs
        // (volatile: two plain 8-byte loads next to each other are merged into ds_read2_b64, which runs at half the
        // rate of ds_read_b64; the positions are only 8-byte aligned, so ds_read_b128 is not available)
        struct f2 { float a, b; };
#ifdef __HIPCC__
        typedef float f2v __attribute__((ext_vector_type(2)));
        typedef const volatile __attribute__((address_space(3))) f2v* lds_f2v;    // (a volatile access through a generic pointer is a flat load)
        const f2v pv = *(lds_f2v)(in + 2 * i);
        const f2 p = {pv.x, pv.y};
#else
        const f2 p = {in[2 * i], in[2 * i + 1]};
#endif
        set_part<IM>(v[2 * i], p.a);
        set_part<IM>(v[2 * i + 1], p.b);
    }
}

// pass 3 after the swap: register 4 d3 + e2lo.  Twiddle (q3 * w64^e2lo)^d3, then 4-point DFTs over d3.
template <int DIR>
SUSHI_HD void w_pass3(cpx* v, const cpx q3) {
    const float s = (float)DIR;
    // exp(DIR 2 pi i e / 64), e = 1, 2, 3
    const cpx w64[4] = {cpx{1.f, 0.f}, cpx{0.99518472667219688624f, s * 0.09801714032956060199f},
                        cpx{0.98078528040323044913f, s * 0.19509032201612826785f},
                        cpx{0.95694033573220886494f, s * 0.29028467725446236764f}};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const cpx z = e == 0 ? q3 : cmul(q3, w64[e]);
        const cpx z2 = cmul(z, z);
        const cpx z3 = cmul(z2, z);
        v[4 + e] = cmul(v[4 + e], z);
        v[8 + e] = cmul(v[8 + e], z2);
        v[12 + e] = cmul(v[12 + e], z3);
        cpx b[4] = {v[e], v[4 + e], v[8 + e], v[12 + e]};
        Dft<4, DIR>::run(b);
        v[e] = b[0]; v[4 + e] = b[1]; v[8 + e] = b[2]; v[12 + e] = b[3];
    }
}

// workgroup exchange: wave n1 holds A_n1[k2] at register 4 e3 + e2lo of lane (e2hi = lane >> 4, e1 = lane & 15),
// k2 = e1 + 64 e2hi + 16 e2lo + 256 e3; it goes to position 18 k2 + n1, thread k2 reads its 16 values back as pairs
template <int IM>
SUSHI_HD void w_wg_store(const cpx* v, int tid, float* lds) {
    const int lane = tid & 63;
    float* out = lds + WROW * ((lane & 15) + 64 * (lane >> 4)) + (tid >> 6);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[WROW * (16 * (r & 3) + 256 * (r >> 2))] = part_of<IM>(v[r]);
}
template <int IM>
SUSHI_HD void w_wg_load(cpx* v, int tid, const float* lds) {
    const float* in = lds + WROW * tid;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // (volatile: two plain 8-byte loads next to each other are merged into ds_read2_b64, which runs at half the
        // rate of ds_read_b64; the positions are only 8-byte aligned, so ds_read_b128 is not available)
        struct f2 { float a, b; };
#ifdef __HIPCC__
        typedef float f2v __attribute__((ext_vector_type(2)));
        typedef const volatile __attribute__((address_space(3))) f2v* lds_f2v;    // (a volatile access through a generic pointer is a flat load)
        const f2v pv = *(lds_f2v)(in + 2 * i);
        const f2 p = {pv.x, pv.y};
#else
        const f2 p = {in[2 * i], in[2 * i + 1]};
#endif
        set_part<IM>(v[2 * i], p.a);
        set_part<IM>(v[2 * i + 1], p.b);
    }
}

#ifdef __HIPCC__
struct NoHook { __device__ __forceinline__ void operator()() const {} };

template <int LOGN, int EX, class Hook = NoHook>
__device__ __forceinline__ void exchange_split(cpx* v, int tid, float* lds, Hook before_last_barrier = Hook()) {
    split_store<LOGN, EX, 0>(v, tid, lds);
    __syncthreads();
    split_load<LOGN, EX, 0>(v, tid, lds);
    __syncthreads();
    split_store<LOGN, EX, 1>(v, tid, lds);
    before_last_barrier();
    __syncthreads();
    split_load<LOGN, EX, 1>(v, tid, lds);
}

// Full transform of the 16 points in v (in_index layout) -> v (out_index layout).  `lds` must hold
// lds_floats<LOGN>() floats; its contents are dead once the call returns and it may be reused after one
// further __syncthreads().  `before_last_pass` runs after the last exchange's stores and before the last
// barrier: a place to issue independent global loads whose latency the last pass covers.
template <int LOGN, int DIR, class Hook = NoHook>
__device__ __forceinline__ void fft_split(cpx* v, int tid, float* lds, const Twiddles tw, Hook before_last_pass = Hook()) {
    typedef Plan<LOGN> P;
    pass_compute<R1, 1, DIR>(v, cpx{1.f, 0.f});
    exchange_split<LOGN, 1>(v, tid, lds);
    pass_compute<R2, R1, DIR>(v, tw.p2);
    __syncthreads();
    exchange_split<LOGN, 2>(v, tid, lds);
    pass_compute<P::R3, R1 * R2, DIR>(v, tw.p3);
    __syncthreads();
    exchange_split<LOGN, 3>(v, tid, lds, before_last_pass);
    pass_compute<R4, P::NT, DIR>(v, tw.p4);
}
// The wave plan's transform of the 16 points in v (register d1 of thread (w, l) = x[wbin(tid, d1)]) -> v[k1] = X[tid + 1024 k1].
// `lds`: W_LDS_FLOATS floats, dead after the call (reusable after one further __syncthreads()).
template <int DIR>
__device__ __forceinline__ void fft_wave(cpx* v, int tid, float* lds, const WTwiddles tw) {
    Dft<16, DIR>::run(v);                                        // pass 1
    // row exchange inside the wave: its LDS operations execute in order, nobody else touches its 1152 floats
    w_row_store<0>(v, tid, lds);
    __builtin_amdgcn_wave_barrier();
    w_row_load<0>(v, tid, lds);
    __builtin_amdgcn_wave_barrier();
    w_row_store<1>(v, tid, lds);
    __builtin_amdgcn_wave_barrier();
    w_row_load<1>(v, tid, lds);
    pass_compute<16, 16, DIR>(v, tw.g2);                         // pass 2 (twiddle, then the DFTs)
    // register bits 3:2 <-> lane bits 5:4
    auto swap32 = [](float& a, float& b) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
    };
    auto swap16 = [](float& a, float& b) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
    };
#pragma unroll
    for (int r = 0; r < 8; ++r) { swap32(v[r].x, v[r + 8].x); swap32(v[r].y, v[r + 8].y); }
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if ((r & 4) == 0) { swap16(v[r].x, v[r + 4].x); swap16(v[r].y, v[r + 4].y); }
    w_pass3<DIR>(v, tw.q3);                                      // pass 3
    __syncthreads();                                             // every wave is done with its row-exchange floats
    w_wg_store<0>(v, tid, lds);
    __syncthreads();
    w_wg_load<0>(v, tid, lds);
    __syncthreads();
    w_wg_store<1>(v, tid, lds);
    __syncthreads();
    w_wg_load<1>(v, tid, lds);
    pass_compute<16, WNT, DIR>(v, tw.p4);                        // pass 4
}
// The same transform with its first pass on the matrix pipe (header of "First pass ... on the matrix pipe" above).  `yl`: the
// thread's four 16-byte entries of a spectrum stored in mslot_of_bin order, as loaded; `b`: the lane's DFT-matrix operands.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
struct MfmaB { half8 rh, rl, ih, il; };
__device__ __forceinline__ MfmaB load_mfma_b(int tid, const uint4v* __restrict__ table) {
    const int lane = tid & 63;
    MfmaB b;
    b.rh = __builtin_bit_cast(half8, table[0 * 64 + lane]);
    b.rl = __builtin_bit_cast(half8, table[1 * 64 + lane]);
    b.ih = __builtin_bit_cast(half8, table[2 * 64 + lane]);
    b.il = __builtin_bit_cast(half8, table[3 * 64 + lane]);
    return b;
}
// Passes 1 - 3 (everything that stays inside a wave): v[4 e3 + e2lo] of lane (e2hi = lane >> 4, e1 = lane & 15) of wave n1 ends
// as A_n1[k2] = sum_n2 x[n1 + 16 n2] w1024^(n2 k2), k2 = e1 + 64 e2hi + 16 e2lo + 256 e3 -- the 1024-point transform of the
// wave's decimated share of the input.  Every output of the whole transform is a sum of sixteen of these, one per wave, times
// unit factors:  |X[k2 + 1024 k1]| <= sum_n1 |A_n1[k2]|  (what bound_kernel uses to leave a block pair untransformed).
template <int DIR>
__device__ __forceinline__ void fft_wave_mfma_front(const uint4v (&yl)[4], cpx* v, int tid, const WTwiddles tw, const MfmaB& b) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        // (re, im) x 4 bins -> re x 8 in the lanes below 32, im x 8 in the lanes above (the other four bins are the partner lane's)
        unsigned rr01 = __builtin_amdgcn_perm(yl[g][1], yl[g][0], 0x05040100u), rr23 = __builtin_amdgcn_perm(yl[g][3], yl[g][2], 0x05040100u);
        unsigned ii01 = __builtin_amdgcn_perm(yl[g][1], yl[g][0], 0x07060302u), ii23 = __builtin_amdgcn_perm(yl[g][3], yl[g][2], 0x07060302u);
        { const auto r = __builtin_amdgcn_permlane32_swap(rr01, ii01, false, false); rr01 = r[0]; ii01 = r[1]; }
        { const auto r = __builtin_amdgcn_permlane32_swap(rr23, ii23, false, false); rr23 = r[0]; ii23 = r[1]; }
        const uint4v aw = {rr01, rr23, ii01, ii23};
        const half8 a = __builtin_bit_cast(half8, aw);
        const float4v z = {0.f, 0.f, 0.f, 0.f};
        float4v dr = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b.rh, z, 0, 0, 0);
        float4v di = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b.ih, z, 0, 0, 0);
        dr = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b.rl, dr, 0, 0, 0);
        di = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b.il, di, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[4 * g + i] = cpx{dr[i], di[i]};
    }
    pass_compute<16, 16, DIR>(v, tw.g2);                         // pass 2 (twiddle, then the DFTs)
    auto swap32 = [](float& x, float& y) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
        x = __uint_as_float(r[0]); y = __uint_as_float(r[1]);
    };
    auto swap16 = [](float& x, float& y) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
        x = __uint_as_float(r[0]); y = __uint_as_float(r[1]);
    };
#pragma unroll
    for (int r = 0; r < 8; ++r) { swap32(v[r].x, v[r + 8].x); swap32(v[r].y, v[r + 8].y); }
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if ((r & 4) == 0) { swap16(v[r].x, v[r + 4].x); swap16(v[r].y, v[r + 4].y); }
    w_pass3<DIR>(v, tw.q3);                                      // pass 3
}
// The workgroup exchange and pass 4.
template <int DIR>
__device__ __forceinline__ void fft_wave_mfma_back(cpx* v, int tid, float* lds, const WTwiddles tw) {
    w_wg_store<0>(v, tid, lds);                                  // (no wave used the buffer before: no barrier in front)
    __syncthreads();
    w_wg_load<0>(v, tid, lds);
    __syncthreads();
    w_wg_store<1>(v, tid, lds);
    __syncthreads();
    w_wg_load<1>(v, tid, lds);
    pass_compute<16, WNT, DIR>(v, tw.p4);                        // pass 4
}
template <int DIR>
__device__ __forceinline__ void fft_wave_mfma(const uint4v (&yl)[4], cpx* v, int tid, float* lds, const WTwiddles tw, const MfmaB& b) {
    fft_wave_mfma_front<DIR>(yl, v, tid, tw, b);
    fft_wave_mfma_back<DIR>(v, tid, lds, tw);
}
// ------------------------------------------------------------------------------------------------------------
// The same three in-wave passes in PACKED HALVES: a complex value is one 32-bit register (re, im), an addition one
// v_pk_add_f16, a multiplication by a twiddle a v_pk_mul_f16 and a v_pk_fma_f16 (the swaps and sign changes ride on the
// instructions' op_sel / neg modifiers), a radix-16 butterfly stage half the instructions of the float32 one.  For
// bound_kernel only, which needs an UPPER BOUND of the largest |A_n1[k2]| to two digits: every output is a sum of 64 of the
// pass-1 values times unit factors through at most 8 roundings of 2^-11 and twiddles rounded to 2^-12, so
//     | |A| computed - |A| | <= (8 * 2^-11 + 2^-11) * 64 * max |pass-1 value| < 0.29 * max |pass-1 value|,
// which the caller adds.  Values are scaled by 2^-10 on the way in (the DFT matrix operand carries the factor): with |Y(f)|
// below 2^15.5 nothing can overflow (16 * 16 * 4 terms: 2^25.5 * 2^-10 < 65504).
// ------------------------------------------------------------------------------------------------------------
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 h_rot(const h2 a) { return h2{-a.y, a.x}; }                            // times +i
__device__ __forceinline__ h2 h_cmul(const h2 a, const h2 w) { return a * w.xx + h_rot(a) * w.yy; }
__device__ __forceinline__ h2 h_splat(const float c) { return h2{(_Float16)c, (_Float16)c}; }
// lo = e + w o, hi = e - w o for w = exp(+2 pi i q / 32) (the inverse transform's roots)
__device__ __forceinline__ void h_bfly_root32(const h2 e, const h2 o, const int q, h2& lo, h2& hi) {
    const float C[9] = {1.0f, 0.98078528040323044913f, 0.92387953251128673848f, 0.83146961230254523708f, 0.70710678118654752440f,
                        0.55557023301960222474f, 0.38268343236508978178f, 0.19509032201612826785f, 0.0f};
    switch (q) {
        case 0: lo = e + o; hi = e - o; return;
        case 8: lo = e + h_rot(o); hi = e - h_rot(o); return;
        case 4: { const h2 p = o + h_rot(o); lo = p * h_splat(C[4]) + e; hi = e - p * h_splat(C[4]); return; }
        case 12: { const h2 p = h_rot(o) - o; lo = p * h_splat(C[4]) + e; hi = e - p * h_splat(C[4]); return; }
        default: {
            const float c = q <= 8 ? C[q] : -C[16 - q], sn = q <= 8 ? C[8 - q] : C[q - 8];
            lo = h_rot(o) * h_splat(sn) + (o * h_splat(c) + e);
            hi = e * h_splat(2.0f) - lo;
        }
    }
}
template <int R>
struct DftH {
    static __device__ __forceinline__ void run(h2* v) {
        h2 e[R / 2], o[R / 2];
#pragma unroll
        for (int u = 0; u < R / 2; ++u) { e[u] = v[2 * u]; o[u] = v[2 * u + 1]; }
        DftH<R / 2>::run(e);
        DftH<R / 2>::run(o);
#pragma unroll
        for (int t = 0; t < R / 2; ++t) h_bfly_root32(e[t], o[t], t * (32 / R), v[t], v[t + R / 2]);
    }
};
template <>
struct DftH<1> {
    static __device__ __forceinline__ void run(h2*) {}
};
// a lane's twiddles of passes 2 and 3 (they depend on the lane only): rounded once from the float32 table
struct HTwiddles { h2 p2[16]; h2 z1[4], z2[4], z3[4]; };
__device__ __forceinline__ h2 h_round(const cpx c) { return h2{(_Float16)c.x, (_Float16)c.y}; }
__device__ __forceinline__ HTwiddles load_htwiddles(const int lane, const cpx* __restrict__ tw) {
    HTwiddles t;
    const int e1 = lane & 15;
#pragma unroll
    for (int d = 0; d < 16; ++d) t.p2[d] = h_round(cconj(tw[((e1 * d) & 255) * (TWIDDLE_N / 256)]));          // w256^(e1 d), inverse
    const cpx q3 = cconj(tw[((lane & 15) + 64 * (lane >> 4)) * (TWIDDLE_N / 1024)]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const cpx z = cmul(q3, cconj(tw[e * (TWIDDLE_N / 64)]));                                                 // q3 w64^e
        const cpx z2 = cmul(z, z);
        t.z1[e] = h_round(z); t.z2[e] = h_round(z2); t.z3[e] = h_round(cmul(z2, z));
    }
    return t;
}
// |a|^2 as float32 bits, and the largest of such: for values >= 0 the order of the bit patterns is the order of the values, and an
// infinity or a NaN (were there ever one) is LARGER than every finite value instead of being dropped, as fmaxf drops a NaN.
// (The builtin, not an asm statement with the instruction's three-operand form: inside bound_kernel's loop -- not in a
// straight-line test -- the asm's result register came back holding the INPUT.)
__device__ __forceinline__ unsigned h_abs2(const h2 a) {
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_fdot2(a, a, 0.f, false));
}
__device__ __forceinline__ unsigned h_max_bits(const unsigned a, const unsigned b) { return a > b ? a : b; }
struct MfmaBh { half8 r, i; };        // the DFT matrix's high halves times 2^-10
__device__ __forceinline__ MfmaBh load_mfma_bh(const int lane, const uint4v* __restrict__ table) {
    MfmaBh b;
    b.r = __builtin_bit_cast(half8, table[lane]);
    b.i = __builtin_bit_cast(half8, table[64 + lane]);
    return b;
}
// v[4 e3 + e2lo] = 2^-10 A_n1[k2] (the float32 version's layout); max_in2 = the largest |pass-1 value|^2 of this lane (float bits)
__device__ __forceinline__ void fft_wave_half_front(const uint4v (&yl)[4], h2* v, const HTwiddles& tw, const MfmaBh& b, unsigned& max_in2,
                                                    h2* pass1_out = nullptr, h2* pass2_out = nullptr, h2* swapped_out = nullptr) {
    unsigned mi = 0u;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        unsigned rr01 = __builtin_amdgcn_perm(yl[g][1], yl[g][0], 0x05040100u), rr23 = __builtin_amdgcn_perm(yl[g][3], yl[g][2], 0x05040100u);
        unsigned ii01 = __builtin_amdgcn_perm(yl[g][1], yl[g][0], 0x07060302u), ii23 = __builtin_amdgcn_perm(yl[g][3], yl[g][2], 0x07060302u);
        { const auto r = __builtin_amdgcn_permlane32_swap(rr01, ii01, false, false); rr01 = r[0]; ii01 = r[1]; }
        { const auto r = __builtin_amdgcn_permlane32_swap(rr23, ii23, false, false); rr23 = r[0]; ii23 = r[1]; }
        const uint4v aw = {rr01, rr23, ii01, ii23};
        const half8 a = __builtin_bit_cast(half8, aw);
        const float4v z = {0.f, 0.f, 0.f, 0.f};
        const float4v dr = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b.r, z, 0, 0, 0);
        const float4v di = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b.i, z, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned packed = __builtin_bit_cast(unsigned, h2{(_Float16)dr[i], (_Float16)di[i]});
            // (seen as two conversions, the value is converted AGAIN, half by half, for every swapped or negated use: opaque from here)
            asm volatile("" : "+v"(packed));
            v[4 * g + i] = __builtin_bit_cast(h2, packed);
            mi = h_max_bits(mi, h_abs2(v[4 * g + i]));
        }
    }
    max_in2 = mi;
    if (pass1_out) {                                                 // (tools/ubench/half_front_check.hip)
#pragma unroll
        for (int t = 0; t < 16; ++t) pass1_out[t] = v[t];
    }
#pragma unroll
    for (int t = 1; t < 16; ++t) v[t] = h_cmul(v[t], tw.p2[t]);     // pass 2: twiddle w256^(e1 d2), then the DFTs over d2
    DftH<16>::run(v);
    if (pass2_out) {
#pragma unroll
        for (int t = 0; t < 16; ++t) pass2_out[t] = v[t];
    }
    // (hipcc 7.2 miscompiles these swaps when their two results are bit-cast straight to half vectors -- the second result
    // becomes a copy of the first: tools/ubench/half_front_check.hip found it -- so both results pass an opaque asm first)
    auto swap32 = [](h2& x, h2& y) {
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
        unsigned ra = r[0], rb = r[1];
        asm volatile("" : "+v"(ra), "+v"(rb));
        x = __builtin_bit_cast(h2, ra); y = __builtin_bit_cast(h2, rb);
    };
    auto swap16 = [](h2& x, h2& y) {
        const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
        unsigned ra = r[0], rb = r[1];
        asm volatile("" : "+v"(ra), "+v"(rb));
        x = __builtin_bit_cast(h2, ra); y = __builtin_bit_cast(h2, rb);
    };
#pragma unroll
    for (int r = 0; r < 8; ++r) swap32(v[r], v[r + 8]);
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if ((r & 4) == 0) swap16(v[r], v[r + 4]);
    if (swapped_out) {
#pragma unroll
        for (int t = 0; t < 16; ++t) swapped_out[t] = v[t];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {                                    // pass 3: twiddle (q3 w64^e2lo)^d3, 4-point DFTs over d3
        h2 q[4] = {v[e], h_cmul(v[4 + e], tw.z1[e]), h_cmul(v[8 + e], tw.z2[e]), h_cmul(v[12 + e], tw.z3[e])};
        DftH<4>::run(q);
        v[e] = q[0]; v[4 + e] = q[1]; v[8 + e] = q[2]; v[12 + e] = q[3];
    }
}
#endif  // __HIPCC__

// ------------------------------------------------------------------------------------------------------------
// The LOW BAND of a spectrum (bins |f| < N/8: LB_BINS = N/4 of them) as its own transform -- bound_low_kernel.
//
// y_low(x) = sum over the band of Y(f) exp(2 pi i f x / N) is a trigonometric polynomial of degree n = N/8.  Sampled at
// M = N/2 equidistant points x = 2 r' its modulus is everywhere at most 1 / cos(pi n / M) = sqrt(2) times the largest sample
// (Ehlich & Zeller 1964; M. Riesz's lemma: a real trigonometric polynomial of degree n and maximum 1 at x* stays above
// cos(n h) at x* + h, |h| <= pi / n -- and a node is never further than pi / M from x*; for complex values rotate the maximum onto
// the real axis).  The samples are an N/2-point inverse transform of
//     V[k] = Y(k) (k < N/8),  Y(k + N/2) (k >= 3N/8),  0 otherwise,
// and, decimated by eight, y_low[2 r'] = sum_g w^(g r') A_g[r' mod 1024] with A_g the 1024-point inverse transform of
// V_g[m] = V[g + 8 m] -- of whose 1024 inputs only m < 256 and m >= 768 exist.  So
//     |y_low(x)| <= sqrt(2) * sum_g max_k |A_g[k]|        for EVERY x,
// from eight half-empty 1024-point transforms per pair instead of the sixteen full ones of the whole spectrum: a sixth of
// the instructions (pass 1 on v_mfma_f32_16x16x16_f16: only the eight non-zero d1 of sixteen enter), a quarter of the bytes.
//
// A group's transform has the digit structure of the wave plan above (m = 64 d1 + 4 d2 + d3, outputs k2 = e1 + 16 e2 + 256 e3,
// the same twiddles): passes 2 and 3 are fft_wave_half_front's own.  Low rows are stored in the order THIS transform loads them:
// the forward transforms end with thread tid holding X[tid + 1024 r] in register r, and the band is r = 0, 1, 14, 15 of every
// thread -- V indices tid + {0, 1024, 6144, 7168}, all in group g = tid & 7, m = t + {0, 128, 768, 896} with t = tid >> 3,
// i.e. d1 = (t >> 6) + {0, 2, 12, 14} of column (d2, d3) = ((t >> 2) & 15, t & 3): ONE 16-byte entry per forward thread.
//   entry position in a low row:  ((g * 4 + g4) * 32 + 16 kq + m'),   kq = t >> 6, g4 = d2 >> 2, m' = (d3 << 2) | (d2 & 3)
//   (group g's 2 KB are contiguous; lane l < 32 of the transforming wave loads entry (g * 4 + g4) * 32 + l for g4 = 0 .. 3)
// A operand of v_mfma_f32_16x16x16_f16, lane (kq' = l >> 4, m' = l & 15): k = 4 kq' + j;  kq' = 0 / 1: real parts of the
// entry with kq = 0 / 1 (d1 = kq + {0, 2, 12, 14}[j]), kq' = 2 / 3: their imaginary parts (one v_permlane32_swap hands them up).
// ------------------------------------------------------------------------------------------------------------
constexpr int LB_HW = WN / 8;                  // the band: bins f < LB_HW and f >= WN - LB_HW
constexpr int LB_BINS = 2 * LB_HW;
constexpr int LB_ENTRIES = LB_BINS / 4;        // 16-byte entries of a low row
constexpr int LB_GROUPS = 8;
SUSHI_HHD int lslot_of_thread(int tid) {
    const int g = tid & 7, t = tid >> 3;
    const int kq = t >> 6, g4 = (t >> 4) & 3, mm = ((t & 3) << 2) | ((t >> 2) & 3);
    return (g * 4 + g4) * 32 + 16 * kq + mm;
}
// the d1 digit of sub-position j of an entry with parity kq, and the V index / frequency bin it is
SUSHI_HHD int lb_d1(int kq, int j) { return kq + (j == 0 ? 0 : (j == 1 ? 2 : (j == 2 ? 12 : 14))); }
SUSHI_HHD int lb_bin_of(int entry, int j) {     // frequency bin (0 .. WN-1) of sub-position j of entry `entry` of a low row
    const int g = entry >> 7, g4 = (entry >> 5) & 3, l = entry & 31, kq = l >> 4, mm = l & 15;
    const int d2 = 4 * g4 + (mm & 3), d3 = mm >> 2;
    const int m = 64 * lb_d1(kq, j) + 4 * d2 + d3;
    const int k = g + 8 * m;                      // index on the N/2 grid
    return k < WN / 8 ? k : k + WN / 2;
}
// ... and the other way: where bin f sits in a low row, as entry * 4 + sub-position; -1 for a bin outside the band
SUSHI_HHD int lslot_of_bin(int f) {
    if (f < 0 || f >= WN || (f >= LB_HW && f < WN - LB_HW)) return -1;
    const int k = f < LB_HW ? f : f - WN / 2;                     // index on the N/2 grid
    const int g = k & 7, m = k >> 3;
    const int d1 = m >> 6, d2 = (m >> 2) & 15, d3 = m & 3;
    const int kq = d1 & 1, b = d1 - kq;                            // d1 = kq + {0, 2, 12, 14}
    const int j = b == 0 ? 0 : (b == 2 ? 1 : (b == 12 ? 2 : 3));
    return (((g * 4 + (d2 >> 2)) * 32 + 16 * kq + ((d3 << 2) | (d2 & 3))) << 2) + j;
}
// B operands of the K = 16 first pass (inverse transform, times 2^-10; high halves only): element j of lane l is row
// k = 4 (l >> 4) + j of column n = l & 15.  form 0: real parts of the result, 1: imaginary parts.
SUSHI_HHD double dft16_low_operand(int form, int l, int j) {
    const int kq = l >> 4, n = l & 15, part = kq >> 1, d1 = lb_d1(kq & 1, j);
    const double PI = 3.14159265358979323846;
    const double wr = __builtin_cos(2.0 * PI * (double)((n * d1) & 15) / 16.0), wi = __builtin_sin(2.0 * PI * (double)((n * d1) & 15) / 16.0);
    return form == 0 ? (part == 0 ? wr : -wi) : (part == 0 ? wi : wr);
}

#ifdef __HIPCC__
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
struct MfmaBl { half4 r, i; };
__device__ __forceinline__ MfmaBl load_mfma_bl(const int lane, const uint2v* __restrict__ table) {
    MfmaBl b;
    b.r = __builtin_bit_cast(half4, table[lane]);
    b.i = __builtin_bit_cast(half4, table[64 + lane]);
    return b;
}
// passes 2 and 3 of the packed-half front (shared by the whole-spectrum and the low-band transforms)
__device__ __forceinline__ void fft_wave_half_tail(h2* v, const HTwiddles& tw) {
#pragma unroll
    for (int t = 1; t < 16; ++t) v[t] = h_cmul(v[t], tw.p2[t]);     // pass 2: twiddle w256^(e1 d2), then the DFTs over d2
    DftH<16>::run(v);
    auto swap32 = [](h2& x, h2& y) {
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
        unsigned ra = r[0], rb = r[1];
        asm volatile("" : "+v"(ra), "+v"(rb));                       // (hipcc 7.2 miscompile of bit-cast swap results: fft_wave_half_front)
        x = __builtin_bit_cast(h2, ra); y = __builtin_bit_cast(h2, rb);
    };
    auto swap16 = [](h2& x, h2& y) {
        const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
        unsigned ra = r[0], rb = r[1];
        asm volatile("" : "+v"(ra), "+v"(rb));
        x = __builtin_bit_cast(h2, ra); y = __builtin_bit_cast(h2, rb);
    };
#pragma unroll
    for (int r = 0; r < 8; ++r) swap32(v[r], v[r + 8]);
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if ((r & 4) == 0) swap16(v[r], v[r + 4]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {                                    // pass 3: twiddle (q3 w64^e2lo)^d3, 4-point DFTs over d3
        h2 q[4] = {v[e], h_cmul(v[4 + e], tw.z1[e]), h_cmul(v[8 + e], tw.z2[e]), h_cmul(v[12 + e], tw.z3[e])};
        DftH<4>::run(q);
        v[e] = q[0]; v[4 + e] = q[1]; v[8 + e] = q[2]; v[12 + e] = q[3];
    }
}
// One group of a low row: yl[g4] = the entry lane (l & 31) loaded for g4 (lanes >= 32 hold a copy of lane l - 32's).
// v[4 e3 + e2lo] = 2^-10 A_g[k2] in fft_wave_half_front's layout; max_in2 = the largest |pass-1 value|^2 of this lane.
__device__ __forceinline__ void fft_wave_half_front_low(const uint4v (&yl)[4], h2* v, const HTwiddles& tw, const MfmaBl& b, unsigned& max_in2,
                                                        h2* pass1_out = nullptr) {
    unsigned mi = 0u;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        unsigned rr01 = __builtin_amdgcn_perm(yl[g][1], yl[g][0], 0x05040100u), rr23 = __builtin_amdgcn_perm(yl[g][3], yl[g][2], 0x05040100u);
        unsigned ii01 = __builtin_amdgcn_perm(yl[g][1], yl[g][0], 0x07060302u), ii23 = __builtin_amdgcn_perm(yl[g][3], yl[g][2], 0x07060302u);
        // the upper lanes take the lower lanes' imaginary parts (v_permlane32_swap: upper half of the first <-> lower half of the second)
        { const auto r = __builtin_amdgcn_permlane32_swap(rr01, ii01, false, false); rr01 = r[0]; ii01 = r[1]; }
        { const auto r = __builtin_amdgcn_permlane32_swap(rr23, ii23, false, false); rr23 = r[0]; ii23 = r[1]; }
        const uint2v aw = {rr01, rr23};
        const half4 a = __builtin_bit_cast(half4, aw);
        const float4v z = {0.f, 0.f, 0.f, 0.f};
        const float4v dr = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b.r, z, 0, 0, 0);
        const float4v di = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b.i, z, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned packed = __builtin_bit_cast(unsigned, h2{(_Float16)dr[i], (_Float16)di[i]});
            asm volatile("" : "+v"(packed));
            v[4 * g + i] = __builtin_bit_cast(h2, packed);
            mi = h_max_bits(mi, h_abs2(v[4 * g + i]));
        }
    }
    max_in2 = mi;
    if (pass1_out) {
#pragma unroll
        for (int t = 0; t < 16; ++t) pass1_out[t] = v[t];
    }
    fft_wave_half_tail(v, tw);
}
#endif  // __HIPCC__

}  // namespace sushi_fft
#endif
