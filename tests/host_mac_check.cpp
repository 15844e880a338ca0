// tests/host_mac_check.cpp -- runs sushi_amd/csrc/mac_core.hpp (the grouped frequency-domain multiply-accumulate
// on the absolute block grid) on the CPU against the plain double-sum definition.  Built and run by
// tests/test_fft_core_host.py; prints the largest relative error over a set of (n_seg, first pair, pairs) shapes.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../sushi_amd/csrc/mac_core.hpp"

using namespace sushi_mac;
typedef std::complex<double> cd;

static float rnd() { return (float)rand() / RAND_MAX - 0.5f; }

// the walk mac_kernel does for one thread: chunks of SMAX segments, groups of SMAX absolute rows
template <int SMAX, int STEP>
static double run(int n_seg, int pair_lo, int npairs, int nz_avail) {
    const int pair_hi = pair_lo + npairs;
    std::vector<c2> T(n_seg), Z(STEP * pair_hi + n_seg + 4 * SMAX + 80), Y(npairs, c2{9e9f, 9e9f, 9e9f, 9e9f});
    for (auto& t : T) t = c2{rnd(), rnd(), rnd(), rnd()};
    for (auto& z : Z) z = c2{rnd(), rnd(), rnd(), rnd()};
    auto lz = [&](long long j) { return j < nz_avail ? Z[j] : zero2(); };     // rows past the stream are zero
    long long jb0, jb1;
    group_range<SMAX, STEP>(pair_lo, pair_hi, &jb0, &jb1);
    for (int c = 0; c * SMAX < n_seg; ++c) {
        c2 tt[SMAX];
        for (int s = 0; s < SMAX; ++s) tt[s] = (c * SMAX + s) < n_seg ? T[c * SMAX + s] : zero2();
        c2 acc[SMAX / STEP];
        for (auto& a : acc) a = c2{7e7f, 7e7f, 7e7f, 7e7f};                   // garbage: every pair must start with mul2
        for (long long jb = jb0; jb <= jb1; jb += SMAX) {
            auto get_z = [&](int u) { return lz(jb + u + (long long)c * SMAX); };
            auto store = [&](int i, bool valid, const c2 v) {
                if (!valid) return;
                if (c == 0) Y[i] = v;
                else { Y[i].ax += v.ax; Y[i].ay += v.ay; Y[i].bx += v.bx; Y[i].by += v.by; }
            };
            mac_group<SMAX, STEP>(jb, pair_lo, pair_hi, tt, acc, get_z, store);
        }
    }
    double worst = 0;
    for (int i = 0; i < npairs; ++i) {
        cd a(0, 0), b(0, 0);
        double mag = 1e-30, magb = 1e-30;
        for (int s = 0; s < n_seg; ++s) {
            const c2 z = lz((long long)STEP * (pair_lo + i) + s);
            a += cd(T[s].ax, T[s].ay) * cd(z.ax, z.ay);
            b += cd(T[s].bx, T[s].by) * cd(z.bx, z.by);
            mag += std::abs(cd(T[s].ax, T[s].ay) * cd(z.ax, z.ay));
            magb += std::abs(cd(T[s].bx, T[s].by) * cd(z.bx, z.by));
        }
        worst = std::fmax(worst, std::abs(a - cd(Y[i].ax, Y[i].ay)) / mag);
        worst = std::fmax(worst, std::abs(b - cd(Y[i].bx, Y[i].by)) / magb);
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
                worst = std::fmax(worst, run<18, 6>(s, lo, p, 1 << 30));
                worst = std::fmax(worst, run<18, 6>(s, lo, p, 6 * (lo + p) - 2));      // stream ends inside the window
                worst = std::fmax(worst, run<4, 2>(s, lo, p, 1 << 30));
                worst = std::fmax(worst, run<8, 2>(s, lo, p, 1 << 30));
                worst = std::fmax(worst, run<12, 2>(s, lo, p, 2 * (lo + p) + 3));
                worst = std::fmax(worst, run<16, 2>(s, lo, p, 1 << 30));
            }
    printf("%.3e\n", worst);
    return worst < 1e-5 ? 0 : 1;
}
