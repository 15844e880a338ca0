// tests/host_mac_check.cpp -- runs sushi_amd/csrc/mac_core.hpp (the grouped frequency-domain multiply-accumulate
// on the absolute block grid, packed-half operands, float32 sums) on the CPU against the plain double-sum definition
// over the SAME half-rounded operands.  Built and run by tests/test_fft_core_host.py; prints the largest relative error
// over a set of (n_seg, first pair, pairs) shapes, then the largest error of the half <-> float helpers.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../sushi_amd/csrc/mac_core.hpp"

using namespace sushi_mac;
typedef std::complex<double> cd;

static float rnd() { return (float)rand() / RAND_MAX - 0.5f; }

// float -> half bits, round to nearest even (normal range and subnormals; no overflow in these tests)
static unsigned to_half(float f) {
    unsigned b;
    memcpy(&b, &f, 4);
    const unsigned s = (b >> 16) & 0x8000u;
    const float a = std::fabs(f);
    if (a == 0.f) return s;
    int e;
    const float m = std::frexp(a, &e);                    // a = m * 2^e, m in [0.5, 1)
    int he = e + 14;                                       // biased exponent of 2 m
    if (he <= 0) {                                         // subnormal: units of 2^-24
        const unsigned q = (unsigned)std::nearbyint(std::ldexp(a, 24));
        return s | q;
    }
    unsigned q = (unsigned)std::nearbyint(std::ldexp(m, 11));   // 11-bit significand, 1024 .. 2048
    if (q == 2048) { q = 1024; ++he; }
    return s | ((unsigned)he << 10) | (q - 1024);
}
static unsigned pack(float re, float im) { return to_half(re) | (to_half(im) << 16); }
static cd unpack(unsigned w) { return cd(half_bits_to_float(w & 0xffffu), half_bits_to_float(w >> 16)); }

// the walk mac_kernel does for one thread: chunks of SMAX segments, groups of SMAX absolute rows
template <int SMAX, int STEP, int ZP = 3>
static double run(int n_seg, int pair_lo, int npairs, int nz_avail) {
    const int pair_hi = pair_lo + npairs;
    std::vector<h8> T(n_seg), Z(STEP * pair_hi + n_seg + 4 * SMAX + 80);
    std::vector<acc4> Y(npairs);
    for (auto& y : Y) for (int k = 0; k < BINS; ++k) y.re[k] = y.im[k] = 9e9f;
    for (auto& t : T) for (int k = 0; k < BINS; ++k) t.w[k] = pack(rnd(), rnd());     // (Re Tt, -Im Tt)
    for (auto& z : Z) for (int k = 0; k < BINS; ++k) z.w[k] = pack(40.f * rnd(), 40.f * rnd());
    auto lz = [&](long long j) { return j < nz_avail ? Z[j] : zero_h8(); };            // rows past the stream are zero
    long long jb0, jb1;
    group_range<SMAX, STEP>(pair_lo, pair_hi, &jb0, &jb1);
    for (int c = 0; c * SMAX < n_seg; ++c) {
        h8 tt[SMAX];
        for (int s = 0; s < SMAX; ++s) tt[s] = (c * SMAX + s) < n_seg ? T[c * SMAX + s] : zero_h8();
        acc4 acc[SMAX / STEP];
        for (auto& a : acc) for (int k = 0; k < BINS; ++k) a.re[k] = a.im[k] = 7e7f;  // garbage: every pair must start with mul4
        for (long long jb = jb0; jb <= jb1; jb += SMAX) {
            auto get_z = [&](int u) { const h8 z = lz(jb + u + (long long)c * SMAX); return zrow{z, rot_mi(z)}; };
            auto store = [&](int i, bool valid, const acc4& v) {
                if (!valid) return;
                if (c == 0) Y[i] = v;
                else for (int k = 0; k < BINS; ++k) { Y[i].re[k] += v.re[k]; Y[i].im[k] += v.im[k]; }
            };
            mac_group<SMAX, STEP, ZP>(jb, pair_lo, pair_hi, tt, acc, get_z, store);
        }
    }
    double worst = 0;
    for (int i = 0; i < npairs; ++i)
        for (int k = 0; k < BINS; ++k) {
            cd a(0, 0);
            double mag = 1e-30;
            for (int s = 0; s < n_seg; ++s) {
                const cd z = unpack(lz((long long)STEP * (pair_lo + i) + s).w[k]);
                const cd tt = std::conj(unpack(T[s].w[k]));                            // Tt = a + ib from the stored (a, -b)
                a += tt * z;
                mag += std::abs(tt * z);
            }
            worst = std::fmax(worst, std::abs(a - cd(Y[i].re[k], Y[i].im[k])) / mag);
        }
    return worst;
}

int main() {
    srand(7);
    double worst = 0;
    const int segs[] = {1, 2, 3, 6, 7, 8, 9, 12, 13, 17, 18, 19, 30, 37};
    const int pairs[] = {1, 2, 3, 5, 8, 9, 44, 118};
    const int los[] = {0, 1, 2, 3, 5, 17, 100};
    for (int s : segs)
        for (int p : pairs)
            for (int lo : los) {
                worst = std::fmax(worst, run<6, 6>(s, lo, p, 1 << 30));
                worst = std::fmax(worst, run<12, 6>(s, lo, p, 1 << 30));
                worst = std::fmax(worst, run<18, 6, 2>(s, lo, p, 1 << 30));
                worst = std::fmax(worst, run<18, 6>(s, lo, p, 6 * (lo + p) - 2));      // stream ends inside the window
                worst = std::fmax(worst, run<30, 6, 2>(s, lo, p, 1 << 30));
                worst = std::fmax(worst, run<4, 2>(s, lo, p, 1 << 30));
                worst = std::fmax(worst, run<8, 2>(s, lo, p, 1 << 30));
                worst = std::fmax(worst, run<12, 2>(s, lo, p, 2 * (lo + p) + 3));
                worst = std::fmax(worst, run<16, 2>(s, lo, p, 1 << 30));
            }
    // the helpers themselves: every half value survives half -> float -> half, and -i z is (im, -re)
    double helper = 0;
    for (unsigned h = 0; h < 0x7c00u; ++h) {
        if (to_half(half_bits_to_float(h)) != h || to_half(half_bits_to_float(h | 0x8000u)) != (h | 0x8000u)) helper = 1;
    }
    {
        const h8 z = {{pack(1.5f, -2.25f), pack(0.f, 3.f), pack(-7.f, 0.125f), pack(6e-6f, -1e-7f)}};
        const h8 r = rot_mi(z);
        for (int k = 0; k < BINS; ++k)
            if (unpack(r.w[k]) != cd(unpack(z.w[k]).imag(), -unpack(z.w[k]).real())) helper = 1;
    }
    printf("%.3e %.1f\n", worst, helper);
    return worst < 1e-5 && helper == 0 ? 0 : 1;
}
