// sushi_amd/csrc/mac_core.hpp -- the per-thread body of the frequency-domain multiply-accumulate
// (sushi_fft.hip mac_kernel), written against callables so tests/host_mac_check.cpp can run it on the CPU.
//
// For four adjacent frequency bins (in the stored order of the spectra) of one search:
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
//
// Operands are PACKED HALVES, products accumulate in float32 on v_dot2_f32_f16 (two multiply-adds per instruction):
// a complex number is one 32-bit word (re | im << 16), four bins are 16 bytes.  With the pattern spectrum stored as
// U = (a, -b) for Tt = a + ib (which is the forward DFT of the segment itself: Tt is its conjugate) and a row Z = c + id
//     Re(Tt Z) = a c - b d = dot2(U, (c,  d))       = dot2(U, Z)
//     Im(Tt Z) = a d + b c = dot2(U, (d, -c))       = dot2(U, -i Z)
// so a complex multiply-accumulate is two instructions once -i Z exists: rot_mi() makes it once per row piece, where
// the rows enter the workgroup's LDS (mac_kernel), not once per search.  (gfx950's dot instructions take neither
// op_sel nor, here, a reason for one.)
#ifndef SUSHI_MAC_CORE_HPP
#define SUSHI_MAC_CORE_HPP

#ifdef __HIPCC__
#define SUSHI_MAC_HD __device__ __forceinline__
#else
#define SUSHI_MAC_HD inline
#include <stdint.h>
#include <string.h>
#endif

namespace sushi_mac {

constexpr int BINS = 4;                       // bins per lane: 16 bytes of a stored spectrum

struct h8 { unsigned w[BINS]; };              // four complex numbers as packed halves, re | im << 16
struct zrow { h8 z, zr; };                    // a row piece Z and -i Z
struct acc4 { float re[BINS], im[BINS]; };    // float32 accumulators of four bins

#ifndef __HIPCC__
// host emulation of the half arithmetic (tests only): exact conversion, products and sum in float
inline float half_bits_to_float(unsigned h) {
    const unsigned s = (h >> 15) & 1u, e = (h >> 10) & 31u, m = h & 1023u;
    float v;
    if (e == 0) v = (float)m * 5.9604644775390625e-8f;                      // subnormal: m * 2^-24
    else if (e == 31) v = m ? __builtin_nanf("") : __builtin_inff();
    else { unsigned bits = ((e + 112u) << 23) | (m << 13); memcpy(&v, &bits, 4); }
    return s ? -v : v;
}
#endif

// a.lo * b.lo + a.hi * b.hi + c, the halves taken as they are
SUSHI_MAC_HD float dot2(unsigned a, unsigned b, float c) {
#ifdef __HIPCC__
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b), c, false);
#else
    return half_bits_to_float(a & 0xffffu) * half_bits_to_float(b & 0xffffu) +
           half_bits_to_float(a >> 16) * half_bits_to_float(b >> 16) + c;
#endif
}

SUSHI_MAC_HD h8 zero_h8() { return h8{{0u, 0u, 0u, 0u}}; }
SUSHI_MAC_HD acc4 zero_acc() { return acc4{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}; }

// -i z for four packed complex numbers: (re, im) -> (im, -re)
SUSHI_MAC_HD h8 rot_mi(const h8 z) {
    h8 r;
#pragma unroll
    for (int k = 0; k < BINS; ++k) r.w[k] = ((z.w[k] >> 16) | (z.w[k] << 16)) ^ 0x80000000u;
    return r;
}

SUSHI_MAC_HD acc4 mul4(const h8 u, const zrow z) {
    acc4 a;
#pragma unroll
    for (int k = 0; k < BINS; ++k) { a.re[k] = dot2(u.w[k], z.z.w[k], 0.f); a.im[k] = dot2(u.w[k], z.zr.w[k], 0.f); }
    return a;
}
SUSHI_MAC_HD void mac4(acc4& a, const h8 u, const zrow z) {
#pragma unroll
    for (int k = 0; k < BINS; ++k) { a.re[k] = dot2(u.w[k], z.z.w[k], a.re[k]); a.im[k] = dot2(u.w[k], z.zr.w[k], a.im[k]); }
}

// first group base (a multiple of SMAX) and number of groups that cover pairs [pair_lo, pair_hi)
template <int SMAX, int STEP>
SUSHI_MAC_HD void group_range(long long pair_lo, long long pair_hi, long long* jb_first, long long* jb_last) {
    *jb_first = (STEP * pair_lo) / SMAX * SMAX;
    *jb_last = (STEP * (pair_hi - 1) + SMAX - 1) / SMAX * SMAX;      // the group holding the last pair's last block
}

// One group: rows jb .. jb + SMAX - 1 (jb a multiple of SMAX).  get_z(u) -> Z_{jb + u} (+ the chunk's shift) with its
// rotation.  store(I - pair_lo, valid, value) is called for EVERY pair that completes in this group -- SMAX / STEP calls
// per group, unconditionally, `valid` telling whether the pair belongs to [pair_lo, pair_hi): the device caller turns an
// invalid one into a store to a dummy line instead of branching around it, so that the number of memory operations per
// group is a compile-time constant (with a conditional store in the loop the compiler cannot count what is in flight and
// drains every outstanding load at every group).
template <int SMAX, int STEP, int ZP = 3, class GetZ, class Store>
SUSHI_MAC_HD void mac_group(const long long jb, const long long pair_lo, const long long pair_hi,
                            const h8 (&tt)[SMAX], acc4 (&acc)[SMAX / STEP], GetZ& get_z, Store& store) {
    static_assert(SMAX % STEP == 0 && SMAX >= STEP, "SMAX must be a multiple of STEP");
    constexpr int RING = SMAX / STEP;
    const long long ib = jb / STEP;                                    // jb is a multiple of SMAX, hence of STEP
    // rows are requested ZP - 1 steps before their use (get_z is an LDS read on the device: its latency then hides
    // behind the multiply-accumulates of one or two rows instead of being waited for at every row)
    zrow zq[ZP];
#pragma unroll
    for (int u = 0; u < ZP - 1 && u < SMAX; ++u) zq[u] = get_z(u);
#pragma unroll
    for (int u = 0; u < SMAX; ++u) {
        const zrow z = zq[u % ZP];
        if (u + ZP - 1 < SMAX) zq[(u + ZP - 1) % ZP] = get_z(u + ZP - 1);
        // Z_{jb+u} belongs to the pair starting at block jb + u - s: same residue mod STEP as u
#pragma unroll
        for (int s = (u % STEP); s < SMAX; s += STEP) {
            const int slot = (((u - s) + SMAX) / STEP) % RING;
            if (s == 0) acc[slot] = mul4(tt[0], z);                    // a new pair starts here
            else mac4(acc[slot], tt[s], z);
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
