// sushi_amd/csrc/mac_core.hpp -- the per-thread body of the frequency-domain multiply-accumulate
// (sushi_fft.hip mac_kernel), written against callables so tests/host_mac_check.cpp can run it on the CPU.
//
// For one pair of adjacent frequency bins:  Y_i = sum_{s < n_seg} Tt_s * Z_{STEP*i+s},  i = 0 .. npairs-1
// where Z_j is the block spectrum at unit offset j of the search window (a unit = one template segment
// length; consecutive block pairs start STEP units apart) and Tt_s the template-segment spectra.
// Z is streamed once; Z_j meets the segments of its own residue (s = j - STEP*i) and feeds a ring of
// SMAX/STEP live outputs; pair i is complete when Z_{STEP*i+SMAX-1} has been consumed.
// Templates with more than SMAX segments are handled SMAX segments at a time, Y accumulating.
//
// Memory pipeline: Z loads are unconditional (the caller's load_z clamps out-of-range blocks to a
// zero block) and issued ZR steps ahead of their use into a register ring, so that ZR loads are in
// flight per thread; groups of SMAX steps in the interior of the stream run without any bounds check
// (CHECK = false), only the first and last groups test which outputs exist.
#ifndef SUSHI_MAC_CORE_HPP
#define SUSHI_MAC_CORE_HPP

#ifdef __HIPCC__
#define SUSHI_MAC_HD __device__ __forceinline__
#else
#define SUSHI_MAC_HD inline
#endif

namespace sushi_mac {

struct c2 { float ax, ay, bx, by; };      // two complex numbers (bins f, f+1)

SUSHI_MAC_HD c2 zero2() { return c2{0.f, 0.f, 0.f, 0.f}; }
SUSHI_MAC_HD float fma_(float a, float b, float c) {
#ifdef __HIPCC__
    return __builtin_fmaf(a, b, c);
#else
    return a * b + c;
#endif
}
SUSHI_MAC_HD c2 mul2(const c2 t, const c2 z) {
    return c2{fma_(-t.ay, z.ay, t.ax * z.ax), fma_(t.ay, z.ax, t.ax * z.ay),
              fma_(-t.by, z.by, t.bx * z.bx), fma_(t.by, z.bx, t.bx * z.by)};
}
SUSHI_MAC_HD void mac2(c2& acc, const c2 t, const c2 z) {
    acc.ax = fma_(-t.ay, z.ay, fma_(t.ax, z.ax, acc.ax));
    acc.ay = fma_(t.ay, z.ax, fma_(t.ax, z.ay, acc.ay));
    acc.bx = fma_(-t.by, z.by, fma_(t.bx, z.bx, acc.bx));
    acc.by = fma_(t.by, z.bx, fma_(t.bx, z.by, acc.by));
}

// register-ring depth of the Z prefetch: a divisor of SMAX so that ring slots are compile-time
template <int SMAX> struct ZRing { static constexpr int value = (SMAX % 8 == 0) ? 8 : ((SMAX % 6 == 0) ? 6 : ((SMAX % 4 == 0) ? 4 : 2)); };

// One group of SMAX consecutive stream steps jr = jb .. jb+SMAX-1 (jb a multiple of SMAX, SMAX a multiple of STEP).
template <int SMAX, int STEP, bool CHECK, bool ACCUM, class LoadZ, class LoadY, class StoreY>
SUSHI_MAC_HD void mac_group(const int jb, const int npairs, const int zoff, const c2 (&tt)[SMAX],
                            c2 (&acc)[SMAX / STEP], c2 (&zbuf)[ZRing<SMAX>::value],
                            LoadZ& load_z, LoadY& load_y, StoreY& store_y) {
    constexpr int RING = SMAX / STEP;
    constexpr int ZR = ZRing<SMAX>::value;
#pragma unroll
    for (int u = 0; u < SMAX; ++u) {
        const int jr = jb + u;
        const c2 z = zbuf[u % ZR];
        zbuf[u % ZR] = load_z(zoff + jr + ZR);                 // used ZR steps from now
        // Z_{jr} belongs to pair i with segment s = jr - STEP*i: same residue mod STEP as jr (jb is a multiple)
#pragma unroll
        for (int s = (u % STEP); s < SMAX; s += STEP) {
            const int slot = (((u - s) + SMAX) / STEP) % RING;
            if (s == 0) acc[slot] = mul2(tt[0], z);            // a new pair starts here
            else mac2(acc[slot], tt[s], z);
        }
        if (u % STEP == STEP - 1) {                             // the pair whose last segment this was
            const int i = (jr - (SMAX - 1)) / STEP;
            const int slot = (((u - (SMAX - 1)) + SMAX) / STEP) % RING;
            if (!CHECK || (jr >= SMAX - 1 && i < npairs)) {
                c2 o = acc[slot];
                if (ACCUM) {
                    const c2 prev = load_y(i);
                    o.ax += prev.ax; o.ay += prev.ay; o.bx += prev.bx; o.by += prev.by;
                }
                store_y(i, o);
            }
        }
    }
}

template <int SMAX, int STEP, bool ACCUM, class LoadZ, class LoadY, class StoreY>
SUSHI_MAC_HD void mac_chunk(const int npairs, const int zoff, const c2 (&tt)[SMAX],
                            LoadZ& load_z, LoadY& load_y, StoreY& store_y) {
    constexpr int ZR = ZRing<SMAX>::value;
    c2 acc[SMAX / STEP];
#pragma unroll
    for (int r = 0; r < SMAX / STEP; ++r) acc[r] = zero2();
    c2 zbuf[ZR];
#pragma unroll
    for (int r = 0; r < ZR; ++r) zbuf[r] = load_z(zoff + r);
    const int total = STEP * (npairs - 1) + SMAX;               // block spectra this chunk consumes
    // The stores of group jb are pairs (jb + STEP - SMAX)/STEP .. (jb + SMAX - STEP)/STEP: all of them exist for
    // SMAX <= jb < interior_end.  Three loops rather than one with a branch in it: the interior loop then has a
    // single body, and the wait counts the compiler derives at its back edge are those of that body (with the
    // checked variant as a second path through the loop it falls back to a full drain at every group).
    const int interior_end = STEP * npairs - SMAX + STEP;
    int jb = 0;
    mac_group<SMAX, STEP, true, ACCUM>(jb, npairs, zoff, tt, acc, zbuf, load_z, load_y, store_y);
    jb += SMAX;
    for (; jb < interior_end; jb += SMAX)
        mac_group<SMAX, STEP, false, ACCUM>(jb, npairs, zoff, tt, acc, zbuf, load_z, load_y, store_y);
    for (; jb < total; jb += SMAX)
        mac_group<SMAX, STEP, true, ACCUM>(jb, npairs, zoff, tt, acc, zbuf, load_z, load_y, store_y);
}

// load_t(s)  -> Tt_s           (0 <= s < n_seg)
// load_z(j)  -> Z_j            (any j >= 0: the callable returns zero past the end of the stream)
// load_y(i), store_y(i, v)     output pair i
template <int SMAX, int STEP, class LoadT, class LoadZ, class LoadY, class StoreY>
SUSHI_MAC_HD void mac_stream(int n_seg, int npairs, LoadT load_t, LoadZ load_z, LoadY load_y, StoreY store_y) {
    static_assert(SMAX % STEP == 0 && SMAX >= STEP, "SMAX must be a multiple of STEP");
    for (int s_lo = 0; s_lo < n_seg; s_lo += SMAX) {
        c2 tt[SMAX];
#pragma unroll
        for (int s = 0; s < SMAX; ++s) tt[s] = (s_lo + s) < n_seg ? load_t(s_lo + s) : zero2();
        if (s_lo == 0) mac_chunk<SMAX, STEP, false>(npairs, s_lo, tt, load_z, load_y, store_y);
        else mac_chunk<SMAX, STEP, true>(npairs, s_lo, tt, load_z, load_y, store_y);
    }
}

}  // namespace sushi_mac
#endif
