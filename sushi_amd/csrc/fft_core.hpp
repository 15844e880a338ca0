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
#else
#define SUSHI_HD inline
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

// exp(DIR * 2*pi*i * q/32), q = 0..15, as compile-time cases (q is a constant after unrolling)
template <int DIR>
SUSHI_HD cpx mul_root32(cpx a, int q) {
    const float H = 0.70710678118654752440f;
    const float C[9] = {1.0f, 0.98078528040323044913f, 0.92387953251128673848f, 0.83146961230254523708f, H,
                        0.55557023301960222474f, 0.38268343236508978178f, 0.19509032201612826785f, 0.0f};
    const float s = (float)DIR;
    switch (q) {
        case 0: return a;
        case 4: return cpx{H * (a.x - s * a.y), H * (a.y + s * a.x)};
        case 8: return cpx{-s * a.y, s * a.x};
        case 12: return cpx{-H * (a.x + s * a.y), H * (s * a.x - a.y)};
        default:
            // cos(2 pi q / 32) = C[q] (q <= 8), -C[16 - q] (q > 8); sin = C[8 - q] (q <= 8), C[q - 8] (q > 8)
            return cmul(a, cpx{q <= 8 ? C[q] : -C[16 - q], s * (q <= 8 ? C[8 - q] : C[q - 8])});
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
        for (int t = 0; t < R / 2; ++t) {
            const cpx ow = mul_root32<DIR>(o[t], t * (32 / R));
            v[t] = cadd(e[t], ow);
            v[t + R / 2] = csub(e[t], ow);
        }
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
#endif  // __HIPCC__

}  // namespace sushi_fft
#endif
