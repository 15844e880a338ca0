// sushi_amd/csrc/fft_core.hpp -- workgroup FFT (complex f32, N = 8192, 512 threads x 16 points) for gfx950.
//
// Used by the overlap-save form of the template match (DESIGN.md "FFT path"): forward transforms of
// destination-stream blocks and template segments, inverse transforms of the per-block products.
// OpenCV's crossCorr() (templmatch.cpp, behind cv2.matchTemplate at reference wav.py:185) is the same
// block-DFT scheme on the CPU.
//
// Stockham autosort, radix plan 8 x 8 x 8 x 16.  A pass of radix R with NS = product of the earlier
// radices processes butterflies j = 0 .. N/R-1:
//     inputs   x[j + t*N/R] * w^(t*k),   k = j mod NS,  w = exp(DIR*2*pi*i / (NS*R)),  t = 0..R-1
//     outputs  y[(j - k)*R + k + t*NS]   = R-point DFT of the inputs
// Pass 1 (NS = 1, no twiddles) takes its inputs from registers, and a thread owns the two ADJACENT
// butterflies j = 2*tid, 2*tid + 1: its inputs x[2*tid + b + 1024*t] are 16-byte pairs, so that the
// caller can fill them with dwordx4 loads (8-byte global loads reach only ~0.6 of the HBM rate on
// gfx950).  In the other passes a thread owns j = tid (+ 512).  The last pass leaves its outputs in
// registers: thread `tid` ends with X[tid + 512*r], r = 0..15.  Between passes the data goes through
// one LDS buffer, padded by one element per 16 (pad(e) = e + e/16) so that the transposing stores are
// (nearly) bank-conflict free; all pad() arithmetic is folded into per-thread bases plus immediates.
//
// The inverse transforms of the hot path use fft8192_split below: real parts, then imaginary parts, through a
// float buffer of half the size with its own conflict-free layout per exchange.
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

constexpr int N = 8192;          // transform length (complex points)
constexpr int LDS_ELEMS = N + N / 16;   // padded buffer, in complex elements

// Two workgroup shapes share the code below:
//   Shape<512>: 512 threads x 16 points, radix plan 8 x 8 x 8 x 16  (three LDS exchanges)
//   Shape<256>: 256 threads x 32 points, radix plan 16 x 16 x 32    (two LDS exchanges, half the waves per barrier)
template <int NT_> struct Shape {
    static constexpr int NT = NT_;
    static constexpr int PER = N / NT_;
};
constexpr int NT = 512;          // the default shape's thread count (forward transforms)
constexpr int PER = N / NT;

struct cpx { float x, y; };

SUSHI_HD cpx cadd(cpx a, cpx b) { return cpx{a.x + b.x, a.y + b.y}; }
SUSHI_HD cpx csub(cpx a, cpx b) { return cpx{a.x - b.x, a.y - b.y}; }
SUSHI_HD cpx cmul(cpx a, cpx b) { return cpx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
SUSHI_HD cpx cconj(cpx a) { return cpx{a.x, -a.y}; }
SUSHI_HD int pad(int e) { return e + (e >> 4); }

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
// butterfly per thread.  tw[n] = exp(-2*pi*i*n/N), n = 0..N-1 (the forward table; the inverse conjugates it).
// Loaded once, before the first barrier, so that no table load sits between two passes.
struct Twiddles { cpx p2, p3, p4; };

template <int NT_, int DIR>
SUSHI_HD Twiddles load_twiddles(int tid, const cpx* __restrict__ tw) {
    Twiddles t;
    if (NT_ == 512) {
        t.p2 = tw[(tid & 7) * (N / 64)];       // pass 2: R = 8,  NS = 8
        t.p3 = tw[(tid & 63) * (N / 512)];     // pass 3: R = 8,  NS = 64
        t.p4 = tw[tid];                        // pass 4: R = 16, NS = 512
    } else {
        t.p2 = tw[(tid & 15) * (N / 256)];     // pass 2: R = 16, NS = 16
        t.p3 = tw[tid];                        // pass 3: R = 32, NS = 256
        t.p4 = cpx{1.f, 0.f};
    }
    if (DIR > 0) { t.p2 = cconj(t.p2); t.p3 = cconj(t.p3); t.p4 = cconj(t.p4); }
    return t;
}

// which butterfly: pass 1 pairs adjacent ones in a thread, the others stride by the workgroup size
template <int NT_, bool FIRST>
SUSHI_HD int butterfly(int tid, int b) { return FIRST ? 2 * tid + b : tid + b * NT_; }

// One pass, register side: twiddle (unless NS == 1) and butterfly the PER points of this thread.
// v[b*R + t] holds input t of the thread's butterfly b; w1 is the base twiddle (shared by all of them).
// The powers w^t are built from the squarings w, w^2, w^4, w^8, w^16 only (each w^t = product of the
// squarings its binary digits select, at most four multiplications deep): five live twiddle registers
// instead of R, which is what lets the wide last pass coexist with everything else in its register budget.
template <int PER_, int R, int NS, int DIR>
SUSHI_HD void pass_compute(cpx* v, const cpx w1) {
    constexpr int NB = PER_ / R;
    if (NS > 1) {
        cpx sq[5];
        sq[0] = w1;
#pragma unroll
        for (int q = 1; q < 5; ++q) sq[q] = cmul(sq[q - 1], sq[q - 1]);
#pragma unroll
        for (int t = 1; t < R; ++t) {
            cpx wt = cpx{1.f, 0.f};
            bool first = true;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                if (t & (1 << q)) {
                    wt = first ? sq[q] : cmul(wt, sq[q]);
                    first = false;
                }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) v[b * R + t] = cmul(v[b * R + t], wt);
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) Dft<R, DIR>::run(v + b * R);
}

// store the outputs of a pass into the (padded) LDS buffer: element base + t*NS with base = (j-k)*R + k.
//   first pass (NS = 1): a thread's butterflies j = 2 tid, 2 tid + 1 fill the PER contiguous elements from
//            PER*tid on: pad(PER tid + u) = PER tid + (PER/16) tid + u + (u >> 4)
//   NS = 8 (R = 8)  : base = 64 (j >> 3) + k, pad(base + 8 t)  = 68 (j >> 3) + k + 8 t + (t >> 1)
//   NS % 16 == 0    : a multiple-of-16 step: pad(base + NS t) = pad(base) + (NS + NS/16) t
template <int NT_, int R, int NS, bool FIRST>
SUSHI_HD void pass_store(const cpx* v, int tid, cpx* lds) {
    constexpr int PER_ = N / NT_;
    constexpr int NB = PER_ / R;
    static_assert(NS == 1 || NS == 8 || NS % 16 == 0, "padding arithmetic");
    static_assert(!FIRST || (NS == 1 && NB == 2), "the first pass pairs two adjacent butterflies per thread");
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = butterfly<NT_, FIRST>(tid, b);
        const int k = j & (NS - 1);
        const int base = (j - k) * R + k;
        if (NS == 1) {
            cpx* out = lds + PER_ * tid + (PER_ / 16) * tid;
#pragma unroll
            for (int t = 0; t < R; ++t) out[(b * R + t) + ((b * R + t) >> 4)] = v[b * R + t];
        } else {
            cpx* out = lds + (NS == 8 ? 68 * (j >> 3) + k : pad(base));
#pragma unroll
            for (int t = 0; t < R; ++t) out[NS == 8 ? 8 * t + (t >> 1) : t * (NS + NS / 16)] = v[b * R + t];
        }
    }
}

// load the inputs of a radix-R pass from the LDS buffer: pad(j + t*N/R) = pad(j) + t*(N/R + N/R/16)
template <int NT_, int R>
SUSHI_HD void pass_load(cpx* v, int tid, const cpx* lds) {
    constexpr int NB = (N / NT_) / R;
    constexpr int STRIDE = N / R;
    static_assert(STRIDE % 16 == 0, "padding arithmetic assumes N/R is a multiple of 16");
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = butterfly<NT_, false>(tid, b);
        const cpx* in = lds + pad(j);
#pragma unroll
        for (int t = 0; t < R; ++t) v[b * R + t] = in[t * (STRIDE + STRIDE / 16)];
    }
}

// Index maps of the whole transform (what the caller needs to load / interpret registers), R1 = the
// first radix (8 for Shape<512>, 16 for Shape<256>):
//   input : v[b*R1 + t] = x[2*tid + b + (N/R1)*t]     b = 0..1, t = 0..R1-1   (adjacent pairs: 16-byte loads)
//   output: v[r]        = X[tid + NT*r]               r = 0..PER-1
template <int NT_>
SUSHI_HD int in_index_t(int tid, int r) {
    constexpr int R1 = (N / NT_) / 2;
    return 2 * tid + (r / R1) + (N / R1) * (r % R1);
}
template <int NT_>
SUSHI_HD int out_index_t(int tid, int r) { return tid + NT_ * r; }
SUSHI_HD int in_index(int tid, int r) { return in_index_t<NT>(tid, r); }
SUSHI_HD int out_index(int tid, int r) { return out_index_t<NT>(tid, r); }

#ifdef __HIPCC__
#define SUSHI_FFT_BARRIER() __syncthreads()
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// Full transform of the PER points in v (in_index layout) -> v (out_index layout).
// `lds` must hold LDS_ELEMS elements; its contents are dead once the call returns and it may be
// reused after one further __syncthreads().  `before_last_pass` runs after the last exchange's stores and
// before the last pass: a place to issue independent global loads whose latency the last pass covers.
template <int DIR, class Hook = NoHook>
__device__ __forceinline__ void fft8192(cpx* v, int tid, cpx* lds, const Twiddles tw, Hook before_last_pass = Hook()) {
    pass_compute<16, 8, 1, DIR>(v, cpx{1.f, 0.f});
    pass_store<512, 8, 1, true>(v, tid, lds);
    SUSHI_FFT_BARRIER();
    pass_load<512, 8>(v, tid, lds);
    pass_compute<16, 8, 8, DIR>(v, tw.p2);
    SUSHI_FFT_BARRIER();
    pass_store<512, 8, 8, false>(v, tid, lds);
    SUSHI_FFT_BARRIER();
    pass_load<512, 8>(v, tid, lds);
    pass_compute<16, 8, 64, DIR>(v, tw.p3);
    SUSHI_FFT_BARRIER();
    pass_store<512, 8, 64, false>(v, tid, lds);
    before_last_pass();
    SUSHI_FFT_BARRIER();
    pass_load<512, 16>(v, tid, lds);
    pass_compute<16, 16, 512, DIR>(v, tw.p4);
}

#endif  // __HIPCC__

// The same transform with the real and imaginary parts exchanged one after the other through `lds`, a buffer
// of SPLIT_LDS_FLOATS floats.  After a part's loads every thread's registers hold the new part next to the OTHER
// part of the old element set, which is then stored in turn.
// A 4-byte element wants other paddings than an 8-byte one (32 banks x 4 bytes, 32 lanes per access), and every
// exchange is free to lay the buffer out its own way as long as its stores and loads agree:
//   after pass 1 : pos(e) = e + (e >> 5)        stores 16 tid + u -> 16 tid + (tid >> 1) + u   (lane stride 16.5: all banks)
//   after pass 2 : pos(e) = e + 8 (e >> 6)      stores 64 a + b + 8 t (tid = 8 a + b) -> 72 a + b + 8 t  (8 (a & 3) + b: all banks)
//   after pass 3 : pos(e) = e                   stores 512 w + l + 64 t: lanes are unit stride as they are
// and the loads, unit stride over the lanes with a constant multiple of 512 between a thread's inputs, are
// conflict free in all three (a 32-lane group never straddles a padding step).  Offsets stay immediates.
constexpr int SPLIT_LDS_FLOATS = N + N / 8;

template <int IM> SUSHI_HD float part_of(const cpx& c) { return IM ? c.y : c.x; }
template <int IM> SUSHI_HD void set_part(cpx& c, const float e) { if (IM) c.y = e; else c.x = e; }

template <int EX, int IM>
SUSHI_HD void split_store(const cpx* v, int tid, float* lds) {
    if (EX == 1) {                                   // pass 1 outputs: the thread's 16 contiguous elements
        float* out = lds + 16 * tid + (tid >> 1);
#pragma unroll
        for (int u = 0; u < 16; ++u) out[u] = part_of<IM>(v[u]);
    } else if (EX == 2) {                            // R = 8, NS = 8: butterflies j = tid, tid + 512
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int j = tid + 512 * b;
            float* out = lds + 72 * (j >> 3) + (j & 7);
#pragma unroll
            for (int t = 0; t < 8; ++t) out[8 * t] = part_of<IM>(v[b * 8 + t]);
        }
    } else {                                         // R = 8, NS = 64: base = 8 (j - k) + k, k = j & 63
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int j = tid + 512 * b;
            float* out = lds + 8 * (j & ~63) + (j & 63);
#pragma unroll
            for (int t = 0; t < 8; ++t) out[64 * t] = part_of<IM>(v[b * 8 + t]);
        }
    }
}

// inputs of the next pass (radix R): x[j + t N/R], j = tid (+ 512)
template <int EX, int R, int IM>
SUSHI_HD void split_load(cpx* v, int tid, const float* lds) {
    constexpr int NB = 16 / R;
    constexpr int S = N / R;                         // 1024 or 512
    constexpr int STEP = EX == 1 ? S + (S >> 5) : (EX == 2 ? S + 8 * (S >> 6) : S);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = tid + 512 * b;
        const float* in = lds + (EX == 1 ? j + (j >> 5) : (EX == 2 ? j + 8 * (j >> 6) : j));
#pragma unroll
        for (int t = 0; t < R; ++t) set_part<IM>(v[b * R + t], in[t * STEP]);
    }
}

#ifdef __HIPCC__
template <int EX, int R_NEXT, class Hook = NoHook>
__device__ __forceinline__ void exchange_split(cpx* v, int tid, float* lds, Hook before_last_barrier = Hook()) {
    split_store<EX, 0>(v, tid, lds);
    SUSHI_FFT_BARRIER();
    split_load<EX, R_NEXT, 0>(v, tid, lds);
    SUSHI_FFT_BARRIER();
    split_store<EX, 1>(v, tid, lds);
    before_last_barrier();
    SUSHI_FFT_BARRIER();
    split_load<EX, R_NEXT, 1>(v, tid, lds);
}

template <int DIR, class Hook = NoHook>
__device__ __forceinline__ void fft8192_split(cpx* v, int tid, float* lds, const Twiddles tw, Hook before_last_pass = Hook()) {
    pass_compute<16, 8, 1, DIR>(v, cpx{1.f, 0.f});
    exchange_split<1, 8>(v, tid, lds);
    pass_compute<16, 8, 8, DIR>(v, tw.p2);
    SUSHI_FFT_BARRIER();
    exchange_split<2, 8>(v, tid, lds);
    pass_compute<16, 8, 64, DIR>(v, tw.p3);
    SUSHI_FFT_BARRIER();
    exchange_split<3, 16>(v, tid, lds, before_last_pass);
    pass_compute<16, 16, 512, DIR>(v, tw.p4);
}

// The same transform by 256 threads x 32 points (Shape<256>).
template <int DIR, class Hook = NoHook>
__device__ __forceinline__ void fft8192_w256(cpx* v, int tid, cpx* lds, const Twiddles tw, Hook before_last_pass = Hook()) {
    pass_compute<32, 16, 1, DIR>(v, cpx{1.f, 0.f});
    pass_store<256, 16, 1, true>(v, tid, lds);
    SUSHI_FFT_BARRIER();
    pass_load<256, 16>(v, tid, lds);
    pass_compute<32, 16, 16, DIR>(v, tw.p2);
    SUSHI_FFT_BARRIER();
    pass_store<256, 16, 16, false>(v, tid, lds);
    before_last_pass();
    SUSHI_FFT_BARRIER();
    pass_load<256, 32>(v, tid, lds);
    pass_compute<32, 32, 256, DIR>(v, tw.p3);
}
#endif

}  // namespace sushi_fft
#endif
