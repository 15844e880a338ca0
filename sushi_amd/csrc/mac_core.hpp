// sushi_amd/csrc/mac_core.hpp -- the per-thread body of the frequency-domain multiply-accumulate
// (sushi_fft.hip mac_kernel), written against callables so tests/host_mac_check.cpp can run it on the CPU.
//
// For one pair of adjacent frequency bins of one search:
//     Y_I = sum_{s < n_seg} Tt_s * Z_{STEP*I + s},      I = pair_lo .. pair_hi - 1
// where Z_j is the block spectrum of ABSOLUTE block j of the destination stream (a block = one pattern-segment
// length; the block pairs of every search start at absolute multiples of STEP) and Tt_s the pattern-segment
// spectra.  Z is consumed in groups of SMAX consecutive absolute blocks jb .. jb + SMAX - 1, jb a multiple of
// SMAX: Z_j meets the segments of its own residue (s = j - STEP*I) and feeds a ring of SMAX/STEP live outputs;
// pair I is complete when Z_{STEP*I + SMAX - 1} has been consumed (segments n_seg .. SMAX-1 are zero).
// Because the grouping is absolute, the searches that share a wave (one per group of eight lanes, mac_kernel)
// walk the same groups together and read every Z row at the same address: a row is fetched from L2 once per
// wave instead of once per search.
// Patterns with more than SMAX segments are handled SMAX segments at a time (Y accumulating): chunk c pairs
// segments c*SMAX + s with blocks STEP*I + c*SMAX + s, i.e. the same walk over rows shifted by c*SMAX.
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

// first group base (a multiple of SMAX) and number of groups that cover pairs [pair_lo, pair_hi)
template <int SMAX, int STEP>
SUSHI_MAC_HD void group_range(long long pair_lo, long long pair_hi, long long* jb_first, long long* jb_last) {
    *jb_first = (STEP * pair_lo) / SMAX * SMAX;
    *jb_last = (STEP * (pair_hi - 1) + SMAX - 1) / SMAX * SMAX;      // the group holding the last pair's last block
}

// One group: rows jb .. jb + SMAX - 1 (jb a multiple of SMAX).  get_z(u) -> Z_{jb + u} (+ the chunk's shift).
// store(I - pair_lo, valid, value) is called for EVERY pair that completes in this group -- SMAX / STEP calls per group,
// unconditionally, `valid` telling whether the pair belongs to [pair_lo, pair_hi): the device caller turns an invalid one
// into a store to a dummy line instead of branching around it, so that the number of memory operations per group is
// a compile-time constant (with a conditional store in the loop the compiler cannot count what is in flight and drains
// every outstanding load at every group).
template <int SMAX, int STEP, int ZP = 3, class GetZ, class Store>
SUSHI_MAC_HD void mac_group(const long long jb, const long long pair_lo, const long long pair_hi,
                            const c2 (&tt)[SMAX], c2 (&acc)[SMAX / STEP], GetZ& get_z, Store& store) {
    static_assert(SMAX % STEP == 0 && SMAX >= STEP, "SMAX must be a multiple of STEP");
    constexpr int RING = SMAX / STEP;
    const long long ib = jb / STEP;                                    // jb is a multiple of SMAX, hence of STEP
    // rows are requested two steps before their use (get_z is an LDS read on the device: its latency then hides
    // behind the multiply-accumulates of two rows instead of being waited for at every row)
    c2 zq[ZP];
#pragma unroll
    for (int u = 0; u < ZP - 1 && u < SMAX; ++u) zq[u] = get_z(u);
#pragma unroll
    for (int u = 0; u < SMAX; ++u) {
        const c2 z = zq[u % ZP];
        if (u + ZP - 1 < SMAX) zq[(u + ZP - 1) % ZP] = get_z(u + ZP - 1);
        // Z_{jb+u} belongs to the pair starting at block jb + u - s: same residue mod STEP as u
#pragma unroll
        for (int s = (u % STEP); s < SMAX; s += STEP) {
            const int slot = (((u - s) + SMAX) / STEP) % RING;
            if (s == 0) acc[slot] = mul2(tt[0], z);                    // a new pair starts here
            else mac2(acc[slot], tt[s], z);
        }
        if (u % STEP == STEP - 1) {                                     // the pair whose last segment this was
            const long long I = ib + (u - (SMAX - 1)) / STEP;          // exact division (possibly negative)
            const int slot = (((u - (SMAX - 1)) + SMAX) / STEP) % RING;
            store((int)(I - pair_lo), I >= pair_lo && I < pair_hi, acc[slot]);
        }
    }
}

}  // namespace sushi_mac
#endif
