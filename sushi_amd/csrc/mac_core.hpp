// sushi_amd/csrc/mac_core.hpp -- the per-thread body of the frequency-domain multiply-accumulate
// (sushi_fft.hip mac_kernel), written against callables so tests/host_mac_check.cpp can run it on the CPU.
//
// For one pair of adjacent frequency bins:  Y_i = sum_{s < n_seg} Tt_s * Z_{2i+s},  i = 0 .. npairs-1
// where Z_j is block j of the search window (j relative to the window's first block) and Tt_s the
// template-segment spectra.  Z is streamed once; Z_j meets the segments of its own parity (s = j - 2i)
// and feeds a ring of SMAX/2 live outputs; pair i is complete when Z_{2i+SMAX-1} has been consumed.
// Templates with more than SMAX segments are handled SMAX segments at a time, Y accumulating.
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
SUSHI_MAC_HD c2 mul2(const c2 t, const c2 z) {
    return c2{t.ax * z.ax - t.ay * z.ay, t.ax * z.ay + t.ay * z.ax, t.bx * z.bx - t.by * z.by, t.bx * z.by + t.by * z.bx};
}
SUSHI_MAC_HD void mac2(c2& acc, const c2 t, const c2 z) {
    acc.ax += t.ax * z.ax - t.ay * z.ay;
    acc.ay += t.ax * z.ay + t.ay * z.ax;
    acc.bx += t.bx * z.bx - t.by * z.by;
    acc.by += t.bx * z.by + t.by * z.bx;
}

// load_t(s)  -> Tt_s           (0 <= s < n_seg)
// load_z(j)  -> Z_j            (j >= 0; the callable returns zero past the end of the stream)
// load_y(i), store_y(i, v)     output pair i
template <int SMAX, class LoadT, class LoadZ, class LoadY, class StoreY>
SUSHI_MAC_HD void mac_stream(int n_seg, int npairs, LoadT load_t, LoadZ load_z, LoadY load_y, StoreY store_y) {
    constexpr int RING = SMAX / 2;
    static_assert(SMAX % 2 == 0 && SMAX >= 2, "SMAX must be even");
    for (int s_lo = 0; s_lo < n_seg; s_lo += SMAX) {
        c2 tt[SMAX];
#pragma unroll
        for (int s = 0; s < SMAX; ++s) tt[s] = (s_lo + s) < n_seg ? load_t(s_lo + s) : zero2();
        c2 acc[RING];
#pragma unroll
        for (int r = 0; r < RING; ++r) acc[r] = zero2();
        const int total = 2 * (npairs - 1) + SMAX;              // block spectra this chunk consumes
        for (int jb = 0; jb < total; jb += SMAX) {
#pragma unroll
            for (int u = 0; u < SMAX; ++u) {
                const int jr = jb + u;
                const c2 z = jr < total ? load_z(s_lo + jr) : zero2();
                // Z_{jr} belongs to pair i with segment s = jr - 2i: same parity as jr (jb is even)
#pragma unroll
                for (int s = (u & 1); s < SMAX; s += 2) {
                    const int slot = (((u - s) + SMAX) / 2) % RING;
                    if (s == 0) acc[slot] = mul2(tt[0], z);      // a new pair starts here
                    else mac2(acc[slot], tt[s], z);
                }
                if (u & 1) {                                      // the pair whose last segment this was
                    const int i = (jr - (SMAX - 1)) / 2;
                    const int slot = (((u - (SMAX - 1)) + SMAX) / 2) % RING;
                    if (jr >= SMAX - 1 && i < npairs) {
                        c2 o = acc[slot];
                        if (s_lo > 0) {
                            const c2 prev = load_y(i);
                            o.ax += prev.ax; o.ay += prev.ay; o.bx += prev.bx; o.by += prev.by;
                        }
                        store_y(i, o);
                    }
                }
            }
        }
    }
}

}  // namespace sushi_mac
#endif
